#!/usr/bin/env python3
"""Profiling aid (CPU): compile one csrc/*.hip to gfx950 assembly and print, per kernel, the static instruction count, barriers,
VGPRs / SGPRs, scratch bytes (a non-zero figure is a spill), LDS and occupancy.

  scripts/asm_stats.py eb_rollout.hip [--filter rollout_tape] [--keep /tmp/x.s] [--scratch-only]

Exit status 1 when --fail-on-scratch is given and any listed kernel has scratch."""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from env_build_amd import build as _b  # noqa: E402


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'], input='\n'.join(names), capture_output=True, text=True).stdout
        return out.split('\n')[:len(names)]
    except OSError:
        return names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('source')
    ap.add_argument('--filter', default='')
    ap.add_argument('--keep', default='')
    ap.add_argument('--scratch-only', action='store_true')
    ap.add_argument('--fail-on-scratch', action='store_true')
    a = ap.parse_args()
    src = a.source if os.path.isabs(a.source) else os.path.join(_b.CSRC, a.source)
    out = a.keep or '/tmp/%s.s' % os.path.splitext(os.path.basename(src))[0]
    if not (a.keep and os.path.isfile(out) and os.path.getmtime(out) > os.path.getmtime(src)):
        subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')] + _b.FLAGS + ['--cuda-device-only', '-S', src, '-o', out],
                              stderr=subprocess.DEVNULL)
    s = open(out).read()
    rows = []
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n', s, re.M):
        end = s.find('.Lfunc_end', m.end())
        nxt = re.search(r'^_Z\w+:', s[end:], re.M)
        tail = s[end:end + (nxt.start() if nxt else 8000)]
        if '; NumVgprs:' not in tail:
            continue
        body = s[m.end():end]

        def g(k):
            mm = re.search(r'; %s: (\d+)' % k, tail)
            return int(mm.group(1)) if mm else -1
        rows.append((m.group(1), len(re.findall(r'^\t[a-z]', body, re.M)), body.count('s_barrier'), g('NumVgprs'), g('NumSgprs'),
                     g('ScratchSize'), g('LDSByteSize'), g('Occupancy')))
    names = demangle([r[0] for r in rows])
    bad = 0
    print('%-110s %6s %4s %5s %5s %7s %7s %4s' % ('kernel', 'inst', 'bar', 'vgpr', 'sgpr', 'scratch', 'lds', 'occ'))
    for r, n in zip(rows, names):
        n = re.sub(r'\(.*', '', n).replace('eb::', '')
        if a.filter and a.filter not in n:
            continue
        if a.scratch_only and r[5] == 0:
            continue
        bad += r[5] > 0
        print('%-110s %6d %4d %5d %5d %7d %7d %4d' % ((n[:110],) + r[1:]))
    return 1 if (a.fail_on_scratch and bad) else 0


if __name__ == '__main__':
    sys.exit(main())
