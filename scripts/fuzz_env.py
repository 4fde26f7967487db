#!/usr/bin/env python3
"""Randomised parity sweep (not part of the test suite): the env-side kernels (eb_env_ego_step, eb_get_obs with and without
exit ids, eb_judge_done, eb_env_step's fused second half through the facade helpers) on the GPU against the CPU oracle."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tests._helpers import DeviceModel, HostModel, oracle_lib
from tests.test_gpu_parity import _random_scene

ap = argparse.ArgumentParser(); ap.add_argument('--seconds', type=float, default=120.0); ap.add_argument('--seed', type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
n = bad = 0
while time.time() < t_end:
    task = ['left', 'straight', 'right'][rng.integers(3)]
    NV = int(rng.choice([1, 3, 5, 8, 9, 16, 32, 64]))
    M = int(rng.choice([1, 2, 7, 16, 23, 40, 64]))
    B = int(rng.choice([1, 63, 64, 65, 300, 1500]))
    nf = int(rng.choice([0, 0, 2]))
    seed = int(rng.integers(1 << 30))
    host, dev = HostModel(oracle_lib(), task, n_veh=NV, n_future=nf), DeviceModel(task, n_veh=NV, n_future=nf)
    ego, cand, cmode, lw, light, act, ref = _random_scene(task, B, M, seed)
    if rng.integers(2):
        cand[:, :, :2] *= np.float32(0.5)
    if rng.integers(4) == 0:
        ego[::5, 3] = rng.uniform(-300, 300, len(ego[::5])).astype(np.float32)     # off the map
    tag = '%s NV=%d M=%d B=%d nf=%d seed=%d' % (task, NV, M, B, nf, seed)
    try:
        (e_h, p_h), (e_d, p_d) = host.env_ego_step(ego, act), dev.env_ego_step(ego, act)
        assert np.array_equal(e_h, e_d) and np.array_equal(p_h, p_d), 'ego step'
        o_h, o_d = host.get_obs(e_h, cand, cmode, light, ref_idx=ref), dev.get_obs(e_h, cand, cmode, light, ref_idx=ref)
        assert np.array_equal(o_h, o_d, equal_nan=True), 'get_obs'
        for lw_ in (lw, None):
            d_h = host.judge_done(e_h, p_h, o_h, cand, cmode, lw_, light)
            d_d = dev.judge_done(e_h, p_h, o_h, cand, cmode, lw_, light)
            assert np.array_equal(d_h, d_d), 'judge_done'
    except AssertionError as e:
        bad += 1
        print('MISMATCH (%s): %s' % (e, tag), flush=True)
    n += 1
print('%d random configurations, %d mismatches' % (n, bad))
