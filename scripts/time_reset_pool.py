#!/usr/bin/env python3
"""Profiling aid: eb_env_reset_pool (the masked reset over the traffic pool, one launch) through the raw C entry, at a given
batch size and mask density.  --tile 0 / 1 / 2 forces 64- / 32- / 16-env tiles (eb_debug_set_tile)."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from env_build_amd.endtoend import CrossroadEnd2end
from env_build_amd.dynamics_and_models import _ptr
ap = argparse.ArgumentParser()
ap.add_argument('--sizes', default='4096,65536'); ap.add_argument('--fracs', default='0.02,0.2,1.0'); ap.add_argument('--iters', type=int, default=200); ap.add_argument('--tile', type=int, default=-1)
a = ap.parse_args()
for B in [int(x) for x in a.sizes.split(',')]:
    env = CrossroadEnd2end('left', n_env=B, multi_display=True)
    env.reset()
    env.api.debug_set_tile(env._h, a.tile)
    for _ in range(3):
        env.step(torch.zeros((B, 2), device=env.device))
    obs2, code2 = torch.empty_like(env._obs), torch.empty_like(env.done_code)
    sp = env._sp()
    for f in [float(x) for x in a.fracs.split(',')]:
        mask = (torch.rand(B, device=env.device) < f).to(torch.uint8) if f < 1.0 else None
        rule = env._reset_rule
        def call(k):
            rule.seed, rule.counter = 7, k
            env.api.env_reset_pool(env._h, env._traffic.h, B, _ptr(mask), C.c_uint64(11), C.c_uint64(k), 1, _ptr(env._ego), _ptr(env._params),
                                   _ptr(env._ref_idx), _ptr(env._virtual), _ptr(env._v_light), _ptr(code2), None, env.n_cand, _ptr(env._cand),
                                   _ptr(env._cand_mode), C.byref(rule), _ptr(obs2), _ptr(env._obs), _ptr(env.done_code), sp)
        for k in range(10): call(k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for k in range(a.iters): call(100 + k)
        e1.record(); torch.cuda.synchronize()
        print('n_env=%6d mask=%.2f tile=%s: %.2f us per reset' % (B, f, {-1: 'auto', 0: 64, 1: 32, 2: 16}[a.tile], e0.elapsed_time(e1) * 1e3 / a.iters), flush=True)
