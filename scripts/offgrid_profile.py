#!/usr/bin/env python3
"""Profiling aid: the vectorised env's 'nobody ever resets' loop (scripts/time_env_step.py regime c) in windows of 100 steps — time per step
next to where the egos are (distance from the junction centre: percentiles, share beyond the 20 m / 400 m grid levels, non-finite) and how
crowded they are (candidates inside the collision test's 10 m box, observation slots inside the reward pairs' 6.364 m, per env).
usage: scripts/offgrid_profile.py [n_env] [auto]   ('auto': the step resets the envs it finishes — the normal loop, for comparison)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from env_build_amd.endtoend import CrossroadEnd2end
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = CrossroadEnd2end('left', n_env=B, multi_display=True, traffic='pool', auto_reset=len(sys.argv) > 2 and sys.argv[2] == 'auto', copy_outputs=False)
env.reset()
act = torch.rand((B, 2), device=env.device) * 2 - 1
for w in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): env.step(act)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    ego = env._ego.cpu().numpy()
    r = np.hypot(ego[:, 3], ego[:, 4])
    fin = np.isfinite(r)
    q = np.percentile(r[fin], [10, 50, 90]) if fin.any() else [np.nan] * 3
    cand = env._cand.cpu().numpy()
    box = ((np.abs(cand[:, :, 0] - ego[:, None, 3]) < 10) & (np.abs(cand[:, :, 1] - ego[:, None, 4]) < 10)).sum(1)
    obs = env._obs.cpu().numpy(); nv = (obs.shape[1] - 9) // 4; veh = obs[:, 9:9 + 4 * nv].reshape(len(obs), nv, 4)
    near = (np.hypot(veh[:, :, 0] - obs[:, None, 3], veh[:, :, 1] - obs[:, None, 4]) < 6.364).sum(1)
    print('steps %4d-%4d: %6.1f us per step; |pos| p10 %.0f p50 %.0f p90 %.0f m; beyond 80 m %.2f, beyond 460 m %.2f, non-finite %.3f; v p50 %.1f; in the 10 m box %.2f per env, slots within 6.364 m %.2f per env'
          % (100 * w, 100 * w + 99, dt * 1e6, q[0], q[1], q[2], (r[fin] > 80).mean(), (r[fin] > 460).mean(), 1 - fin.mean(), np.median(ego[fin, 0]), box.mean(), near.mean()))
# where the time is: device time of 200 more steps (events on the launch stream) next to their wall time, and the step kernel's phase marks
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
for _ in range(200): env.step(act)
e1.record(); torch.cuda.synchronize()
print('200 more steps: wall %.1f us per step, device %.1f us per step' % ((time.perf_counter() - t0) / 200 * 1e6, e0.elapsed_time(e1) / 200 * 1e3))
import ctypes as C
nb = (B + 63) // 64
tr = torch.zeros((nb * 8, 16), dtype=torch.int64, device=env.device)
env.api.debug_set_trace(env._h, C.c_void_p(tr.data_ptr()), tr.numel()); env.step(act); torch.cuda.synchronize(); env.api.debug_set_trace(env._h, None, 0)
t = tr.cpu().numpy().astype(np.float64)[:nb * 4]; raw = t.copy(); t = (t - t[:, 0][t[:, 0] > 0].min()) / 100.0
for w in range(4):
    rows = t[w::4]
    def d(a, b):
        sel = (raw[w::4, a] > 0) & (raw[w::4, b] > 0)
        return ' '.join('%5.2f' % v for v in np.percentile(rows[sel, a] - rows[sel, b], [10, 50, 90, 99])) if sel.any() else '-'
    print('wave %d: phase 1 %s | barrier 1 wait %s | %s %s | to barrier 3 %s | rows %s   (p10 p50 p90 p99, us)'
          % (w, d(1, 0), d(8, 1), 'tracking' if w == 0 else 'pairs/coll', d(9, 8) if w == 0 else d(2, 8), d(10, 8), d(4, 10)))
print('kernel: first wave start .. last rows stored %.2f us' % (t[:, 4].max()))
# which positions make a tile's tracking slow: the egos of the slowest tiles (the state the marked launch started from is one step old)
w0 = t[0::4]; dur = np.where((raw[0::4, 9] > 0) & (raw[0::4, 8] > 0), w0[:, 9] - w0[:, 8], 0)
ref = env._ref_idx.cpu().numpy()
for b in np.argsort(-dur)[:6]:
    e = ego[64 * b:64 * b + 64]; rr = np.hypot(e[:, 3], e[:, 4])
    far = np.argsort(-rr)[:3]
    print('tile %d: tracking %.1f us; farthest egos: %s' % (b, dur[b], '; '.join('(%.0f, %.0f) path %d v %.1f' % (e[k, 3], e[k, 4], ref[64 * b + k], e[k, 0]) for k in far)))
