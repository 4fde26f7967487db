#!/usr/bin/env python3
"""Profiling aid (GPU box only, on the scratch copy gpurun makes): where the one-launch env step spends its time.  Each variant
patches ONE stage out of csrc/eb_env_step_body.h in place (results are wrong, timings are what is wanted), rebuilds the library and
runs `bench.py --env-step`; the source is restored at the end.  Usage: python scripts/ablate_env_step.py [--flows] [variant ...]
(--flows: the facade's step over the flow source, 65 536 envs x 60 candidates, scripts/time_env_step.py, instead)."""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'env_build_amd', 'csrc', 'eb_env_step_body.h')
VARIANTS = {
    'base': [],
    'no_predict': [("            } else if (OBS) s_cand[e * RS4 + c] = v;", "            } else if (OBS || true) s_cand[e * RS4 + c] = v;")],     # (and no candidate store)
    'no_reward_pairs': [("            for (int base = 0; base < n_pairs; base += 64 * 4) {", "            for (int base = 0; base < 0; base += 64 * 4) {")],
    'no_collision': [("            for (int base = 0; base < n_rec; base += 128 * 4) {", "            for (int base = 0; base < 0; base += 128 * 4) {")],
    'no_slots': [("        const int n_first = __popcll(A.first_mask);", "        const int n_first = 0;")],
    'no_tracking': [("            if (p < 0) { for (int c = 0; c < T; ++c) orow[6 + c] = 0.0f; }", "            if (true) { for (int c = 0; c < T; ++c) orow[6 + c] = 0.0f; }")],
    'no_cand_store': [("                else reinterpret_cast<float4*>(A.cand)[(size_t)e0 * m_cand + idx] = o;", "                else if (e0 < 0) reinterpret_cast<float4*>(A.cand)[(size_t)e0 * m_cand + idx] = o;")],
    'no_row_store': [("        for (int base = tid; base < total; base += 4 * NT) {", "        for (int base = tid; base < 0; base += 4 * NT) {")],
    'no_ego_roles': [("    } else if (wave < 2 && live) {", "    } else if (false) {")],
    'no_reward_sums': [("    if (!OBS && wave == 1 && live) {\n        // E2E:134", "    if (false) {\n        // E2E:134")],
    'no_judge_bits': [("        s_jb[lane] = live ? (uint8_t)judge_bits(TASK, eg.w, s_r[lane], eg.x, eg.y, eg.z, s_miu[lane], red_light) : (uint8_t)0xff;",
                       "        s_jb[lane] = live ? (uint8_t)15 : (uint8_t)0xff;")],
}
orig = open(SRC).read()
FLOWS = '--flows' in sys.argv
names = [x for x in sys.argv[1:] if x != '--flows'] or list(VARIANTS)
out = {}
try:
    for n in names:
        s = orig
        for a, b in VARIANTS[n]:
            assert s.count(a) == 1, (n, a)
            s = s.replace(a, b)
        open(SRC, 'w').write(s)
        if FLOWS:
            r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'time_env_step.py'), '--traffic', 'flows', '--sizes', '65536', '--steps', '60'],
                               capture_output=True, text=True)
            lines = [m for m in (re.search(r'no resets, copy_outputs=False: ([0-9.]+) us', l) for l in r.stdout.splitlines()) if m]
            out[n] = {65536: float(lines[0].group(1))} if lines else {}
        else:
            r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--env-step'], capture_output=True, text=True)
            lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
            out[n] = {l['n_env_per_gpu']: l['avg_launch_us'] for l in lines}
        print('%-16s %s' % (n, '  '.join('%6d envs: %6.2f us' % kv for kv in sorted(out[n].items()))) if lines else '%s FAILED: %s' % (n, r.stderr[-400:]), flush=True)
finally:
    open(SRC, 'w').write(orig)
if 'base' in out:
    for n, v in out.items():
        if n != 'base' and v:
            print('%-16s saves %s' % (n, '  '.join('%6d: %5.2f us' % (k, out['base'][k] - v[k]) for k in sorted(v))))
