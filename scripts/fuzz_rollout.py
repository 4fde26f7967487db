#!/usr/bin/env python3
"""Randomised parity sweep (not part of the test suite): eb_rollout_step / _tape / _gated on the GPU against the CPU
oracle over random tasks, slot counts, look-ahead counts, modes, batch sizes, tile shapes and storage types."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from env_build_amd.synthetic import assemble_obs, make_rollout_inputs
from tests._helpers import DeviceModel, HostModel, oracle_lib

ap = argparse.ArgumentParser(); ap.add_argument('--seconds', type=float, default=120.0); ap.add_argument('--seed', type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
n = bad = 0
while time.time() < t_end:
    task = ['left', 'straight', 'right'][rng.integers(3)]
    N = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 31, 32, 33, 48, 63, 64]))
    nf = int(rng.choice([0, 0, 0, 1, 3]))
    mode = ['training', 'selecting'][rng.integers(2)]
    B = int(rng.choice([1, 2, 63, 64, 65, 200, 1000, 2049, 5000]))
    H = int(rng.integers(1, 5))
    tile = int(rng.choice([-1, 0, 1, 2]))
    f16 = bool(rng.integers(4) == 0)
    seed = int(rng.integers(1 << 30))
    host, dev = HostModel(oracle_lib(), task, n_veh=N, n_future=nf, mode=mode), DeviceModel(task, n_veh=N, n_future=nf, mode=mode)
    dev.set_tile(tile)
    sched = (int(rng.choice([-1, 0, 1])), int(rng.choice([-1, 0, 1])))   # eb_debug_set_rollout_sched: rolling loads / priority by progress
    dev.set_rollout_sched(*sched)
    inp = make_rollout_inputs(task, B, N, H, seed=seed, n_future=nf)
    if mode == 'selecting':
        inp['ref_idx'][:] = int(rng.integers(3))
    if rng.integers(3) == 0:                       # crowd some envs
        veh = inp['veh'].reshape(B, N, 4)
        k = max(1, B // 4)
        veh[:k, :, 0] = inp['ego'][:k, None, 3] + rng.uniform(-5, 5, (k, N)).astype(np.float32)
        veh[:k, :, 1] = inp['ego'][:k, None, 4] + rng.uniform(-5, 5, (k, N)).astype(np.float32)
        inp['veh'] = veh.reshape(B, 4 * N)
    ego = inp['ego']
    trk = host.tracking_error(ego[:, 3], ego[:, 4], ego[:, 5], ego[:, 0], nf, ref_idx=inp['ref_idx'])
    obs0 = assemble_obs(ego, trk, inp['veh'])
    ri = inp['ref_idx'] if mode == 'training' else None
    pid = 0 if mode == 'training' else int(inp['ref_idx'][0])
    tag = '%s N=%d nf=%d %s B=%d H=%d tile=%d sched=%s f16=%d seed=%d' % (task, N, nf, mode, B, H, tile, sched, f16, seed)
    try:
        if f16:
            o16 = obs0.astype(np.float16).view(np.uint16)
            a_o, a_5 = dev.rollout_tape_f16(o16, inp['actions'], ri, path_id=pid)
            b_o, b_5 = host.rollout_tape_f16(o16, inp['actions'], ri, path_id=pid)
        else:
            a_o, a_5 = dev.rollout_tape(obs0, inp['actions'], ri, path_id=pid)
            b_o, b_5 = host.rollout_tape(obs0, inp['actions'], ri, path_id=pid)
            oh = od = obs0
            for t in range(H):                     # and step by step
                oh, o5h, _ = host.rollout_step(oh, inp['actions'][t], ri, path_id=pid)
                od, o5d, _ = dev.rollout_step(od, inp['actions'][t], ri, path_id=pid)
                assert np.array_equal(oh, od), 'step obs'
                assert np.array_equal(o5h[[0, 4]], o5d[[0, 4]]) and np.allclose(o5h, o5d, rtol=1e-6, atol=0), 'step out5'
            assert np.array_equal(od, a_o), 'tape vs stepwise'
        assert np.array_equal(a_o, b_o), 'tape obs'
        assert np.array_equal(a_5[:, [0, 4]], b_5[:, [0, 4]]) and np.allclose(a_5, b_5, rtol=1e-6, atol=0), 'tape out5'
    except AssertionError as e:
        bad += 1
        print('MISMATCH (%s): %s' % (e, tag), flush=True)
    n += 1
print('%d random configurations, %d mismatches' % (n, bad))
