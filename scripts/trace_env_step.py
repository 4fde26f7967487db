#!/usr/bin/env python3
"""Profiling aid: per-wave phase timeline of ONE eb_env_step launch (wall-clock marks written by env_step_kernel)."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from env_build_amd import _capi
from env_build_amd.endtoend import CrossroadEnd2end
ap = argparse.ArgumentParser(); ap.add_argument('--n-env', type=int, default=65536); ap.add_argument('--n-cand', type=int, default=16)
a = ap.parse_args()
B = a.n_env
env = CrossroadEnd2end('left', n_env=B, multi_display=True, traffic='pool', n_cand=a.n_cand)
env.seed(0); env.reset()
act = (torch.rand((B, 2), device=env.device) * 0.6 - 0.3).contiguous()
lib = env.api.lib
lib.eb_debug_set_trace.argtypes = [C.c_void_p, C.c_void_p]
for _ in range(3): env.step(act)
torch.cuda.synchronize()
te = 16 if B <= 6144 else 32 if B <= 24576 else 64
nb = (B + te - 1) // te
trs = [torch.zeros((nb * 4, 8), dtype=torch.int64, device=env.device) for _ in range(3)]
for k in range(3):
    lib.eb_debug_set_trace(env._h, C.c_void_p(trs[k].data_ptr())); env.step(act)
torch.cuda.synchronize(); lib.eb_debug_set_trace(env._h, None)
sp = []
for x in trs:
    x = x.cpu().numpy(); sp.append((x[:, 0].min(), x[:, 4].max()))
for k in (1, 2):
    print('launch %d: first wave start .. last wave end %.2f us; dead time since the previous launch %.2f us' % (k, (sp[k][1] - sp[k][0]) / 100., (sp[k][0] - sp[k - 1][1]) / 100.))
t = trs[1].cpu().numpy().astype(np.float64); t = (t - t[:, 0].min()) / 100.0
# marks: 0 start | 1 phase 1 done | 5 (waves 1-3) reward pairs done | 2 phase 2 done | 6 (after the barrier; wave 1: sums + done predicates) | 3 phase 3 done | 4 end
order = [(0, 'start'), (1, 'phase 1 done: ego step / tyre params / traffic step (before barrier 1)'), (5, 'phase 2a: reward pairs done (waves 1-3)'),
         (2, 'phase 2 done: tracking | tags + collision (before barrier 2)'), (6, 'after barrier 2 (wave 1: + penalty sums, done predicates)'),
         (7, 'phase 3: candidate set of the (last) owned mode built'), (3, 'phase 3 done: slots built (before barrier 3)'), (4, 'end: done code, rows and candidates stored')]
def q(x): return ' '.join('%6.2f' % v for v in np.percentile(x, [0, 10, 50, 90, 100]))
print('%d blocks; us since the first wave started; percentiles 0 10 50 90 100' % nb)
for w in range(4):
    print('wave %d' % w)
    for k, n in order:
        if k == 5 and w == 0: continue
        print('  %-86s %s' % (n, q(t[w::4, k])))
