#!/usr/bin/env python3
"""A/B aid (scripts/r5_ab_session.sh): bench.py's env-step entries and the facade's no-resets loop on the library named by EB_AB_LIB
(empty: the in-tree one)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from env_build_amd import _capi
if os.environ.get('EB_AB_LIB'):
    _capi.PROTOTYPES.pop('eb_debug_check_grids', None)      # (an entry the older library does not have; nothing here calls it)
    _capi._hip_api = _capi.CApi(os.environ['EB_AB_LIB'])
import bench, io, contextlib
sys.argv = ['bench.py', '--env-step']
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        bench.main()
    except SystemExit:
        pass
out = buf.getvalue()
for l in out.splitlines():
    try:
        d = json.loads(l)
    except ValueError:
        continue
    a = d['step_with_auto_reset']
    print('%-46s step %.2f us  step + auto reset %.2f us' % (d['workload'][:46], d['avg_launch_us'], a['us_per_step']))
from env_build_amd.endtoend import CrossroadEnd2end
for B in (4096, 65536):
    env = CrossroadEnd2end('left', n_env=B, multi_display=True, traffic='pool', auto_reset=False, copy_outputs=False)
    env.reset()
    act = torch.rand((B, 2), device=env.device) * 2 - 1
    for _ in range(200): env.step(act)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(1000): env.step(act)
    torch.cuda.synchronize()
    print('facade, nobody ever resets, %6d envs: %.1f us per step' % (B, (time.perf_counter() - t0) / 1000 * 1e6))
    del env
