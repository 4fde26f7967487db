#!/usr/bin/env python3
"""Profiling aid: per-wave phase timeline of ONE eb_env_step launch (wall-clock marks written by env_step_kernel)."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from env_build_amd import _capi
from env_build_amd.endtoend import CrossroadEnd2end
ap = argparse.ArgumentParser(); ap.add_argument('--n-env', type=int, default=65536); ap.add_argument('--n-cand', type=int, default=16); ap.add_argument('--auto', action='store_true'); ap.add_argument('--flows', action='store_true', help='the flow traffic source (60 candidates) instead of the pool'); ap.add_argument('--waves', type=int, default=0, help='eb_debug_set_env_waves: 4 / 8 waves per block (0: by grid size)'); ap.add_argument('--wild', action='store_true', help='full-range random actions: many envs finish per step (the reset tail runs in most tiles)')
a = ap.parse_args()
B = a.n_env
env = CrossroadEnd2end('left', n_env=B, multi_display=True, traffic='flows' if a.flows else 'pool', n_cand=None if a.flows else a.n_cand, auto_reset=a.auto, copy_outputs=False)
env.seed(0); env.reset()
act = (torch.rand((B, 2), device=env.device) * (2.0 if a.wild else 0.6) - (1.0 if a.wild else 0.3)).contiguous()
lib = env.api.lib
env.api.debug_set_env_waves(env._h, a.waves)
for _ in range(40 if a.wild else 3): env.step(act)
torch.cuda.synchronize()
te = 16 if (B <= 1024 or a.flows) else 32 if B <= 20480 else 64
nb = (B + te - 1) // te
# rows of 16 words per wave, 4 or 8 waves per block (by grid size, csrc/eb_env_step.hip: launch_env_step): sized for 8, the count is
# read off the marks (wave rows 4-7 of a four-wave launch stay zero)
trs = [torch.zeros((nb * 8, 16), dtype=torch.int64, device=env.device) for _ in range(3)]
for k in range(3):
    env.api.debug_set_trace(env._h, C.c_void_p(trs[k].data_ptr()), trs[k].numel()); env.step(act)
torch.cuda.synchronize(); env.api.debug_set_trace(env._h, None, 0)
nw = 8 if int((trs[1][:, 0] != 0).sum().item()) > nb * 4 else 4
trs = [x[:nb * nw] for x in trs]
sp = []
for x in trs:
    x = x.cpu().numpy(); sp.append((x[:, 0].min(), max(x[:, 4].max(), x[:, 15].max())))
for k in (1, 2):
    print('launch %d: first wave start .. last wave end %.2f us; dead time since the previous launch %.2f us' % (k, (sp[k][1] - sp[k][0]) / 100., (sp[k][0] - sp[k - 1][1]) / 100.))
t = trs[1].cpu().numpy().astype(np.float64); t = (t - t[:, 0].min()) / 100.0
# marks (this wave's lane 0): see `order`
order = [(0, 'start'), (7, 'flows: first chunk staged') if a.flows else (0, 'start'), (11, 'flows: first group staged') if a.flows else (0, 'start'), (1, 'phase 1 done: ego step / tyre params / traffic step (before barrier 1)'), (8, 'after barrier 1'),
         (9, 'wave 0: closest point + tracking done (before the candidate store)'), (5, 'phase 2a: reward pairs done (waves 1-3)'),
         (2, 'phase 2 done: tracking + cand store | pairs + collision (before barrier 2)'), (6, 'after barrier 2 + sums (wave 1) / done code (wave 0)'),
         (7, 'phase 3: candidate set built (last owned mode)') if not a.flows else (0, 'start'), (11, 'phase 3: walk done (no --auto)') if not a.flows else (0, 'start'), (12, 'phase 3: slots written (no --auto)'), (3, 'phase 3 done: slots built (before barrier 3)'), (10, 'after barrier 3'), (4, 'rows stored (end of the step proper)'),
         (11, 'auto: after the drain + barrier'), (12, 'auto: draws done, after the barrier'), (13, 'auto: pool re-entry done (before the barrier)'),
         (14, 'auto: tracking + slots done (before the barrier)'), (15, 'auto: rows stored, flags swapped (end)')]
def q(x): return ' '.join('%6.2f' % v for v in np.percentile(x, [0, 10, 50, 90, 100])) + '   n=%d' % len(x)
print('%d blocks; us since the first wave started; percentiles 0 10 50 90 100' % nb)
raw = trs[1].cpu().numpy()
for w in range(nw):
    print('wave %d' % w)
    for k, n in order:
        sel = raw[w::nw, k] != 0
        if not sel.any(): continue
        print('  %-86s %s' % (n, q(t[w::nw, k][sel])))
# per-block durations: each mark minus the same wave's previous mark (in the order above), percentiles over the blocks — what a phase
# costs a wave, free of when its block happened to start
print('per-block phase durations (us): this mark minus the wave\'s previous one; percentiles 10 50 90')
for w in range(nw):
    print('wave %d' % w)
    prev = None
    for k, n in order[:15]:
        sel = raw[w::nw, k] != 0
        if not sel.any() or (sel.sum() < nb // 2): continue
        if prev is not None:
            both = sel & (raw[w::nw, prev] != 0)
            d = t[w::nw, k][both] - t[w::nw, prev][both]
            print('  %-86s %s' % (n, ' '.join('%6.2f' % v for v in np.percentile(d, [10, 50, 90]))))
        prev = k
blk = t.reshape(nb, nw, 16)
life = blk[:, :, 4].max(1) - blk[:, :, 0].min(1)
print('tile lifetime (first wave start .. last wave rows stored): ' + ' '.join('%6.2f' % v for v in np.percentile(life, [10, 50, 90])))
