#!/usr/bin/env python3
"""Randomised parity sweep (not part of the test suite): eb_env_step — the one-launch kernel, both tile shapes, random
slot / candidate counts, arbitrary candidate modes, per-candidate sizes, re-entry rule — on the GPU against the CPU oracle's
composite, eb_get_obs with a row mask, and eb_env_reset_pool (the one-launch masked reset, with and without carry-over)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from env_build_amd import _capi
from tests._helpers import DeviceModel, HostModel, oracle_lib
from tests._env_step_check import random_scene

ap = argparse.ArgumentParser(); ap.add_argument('--seconds', type=float, default=120.0); ap.add_argument('--seed', type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
n = bad = 0
while time.time() < t_end:
    task = ['left', 'straight', 'right'][rng.integers(3)]
    NV = [None, 1, 3, 16, 32][rng.integers(5)]
    M = int(rng.choice([1, 2, 7, 16, 23, 40, 60, 64]))
    B = int(rng.choice([1, 15, 16, 17, 63, 64, 65, 300, 1500]))
    nf = int(rng.choice([0, 0, 2]))
    tile = int(rng.choice([-1, 0, 1, 2]))
    seed = int(rng.integers(1 << 30))
    kw = dict(mode='training', n_future=nf)
    if NV is not None:
        kw['n_veh'] = NV
    tmodes = [_capi.VMODES[k] for k in rng.integers(0, 12, M)]
    tag = '%s NV=%s M=%d B=%d nf=%d tile=%d seed=%d' % (task, NV, M, B, nf, tile, seed)
    try:
        ego, cand, cmode, lw, light, act, ref = random_scene(task, B, M, seed)
        if rng.integers(2):
            cand[:, :, :2] *= np.float32(0.5)
        if rng.integers(4) == 0:
            ego[::5, 3] = rng.uniform(-300, 300, len(ego[::5])).astype(np.float32)     # off the map
        raw = rng.uniform(-1.2, 1.2, (B, 2)).astype(np.float32)
        virtual = (rng.random(B) < 0.3).astype(np.uint8)
        v_light = rng.integers(0, 4, B).astype(np.uint8)
        entry = rng.uniform(-60, 60, (M, 5)).astype(np.float32)
        rule = dict(entry=entry, limit=float(rng.choice([30.0, 65.0])), span=60.0, v_max=8.0, seed=int(rng.integers(1 << 40)), counter=int(rng.integers(1 << 20))) if rng.integers(3) else None
        outs = []
        for make in (lambda t, **k: HostModel(oracle_lib(), t, **k), lambda t, **k: DeviceModel(t, **k)):
            m, tr = make(task, **kw), make(task, n_veh=M, modes=tmodes)
            if hasattr(m, 'set_tile'):
                m.set_tile(tile)
            obs0 = m.get_obs(ego, cand, cmode, light, ref_idx=ref)
            got = m.env_step(tr, obs0, raw, ego, cand, cmode, ref_idx=ref, cand_lw=lw if rng.integers(2) or True else None,
                             v_light=v_light, virtual=virtual, respawn=rule)
            mask = (np.random.default_rng(seed).random(B) < 0.3)
            init = np.full(obs0.shape, 7.0, np.float32)
            masked = m.get_obs(ego, cand, cmode, light, ref_idx=ref, row_mask=mask.astype(np.uint8), obs_init=init)
            # the masked reset as one launch (eb_env_reset_pool): random mask density, carry-over from other arrays or in place
            rs = np.random.default_rng(seed + 1)
            dens = float(rs.choice([0.02, 0.3, 1.0]))
            rmask = None if dens == 1.0 and rs.integers(2) else (rs.random(B) < dens).astype(np.uint8)
            pool = dict(entry=entry, span=60.0, v_max=8.0, seed=int(rs.integers(1 << 40)), counter=int(rs.integers(1 << 20)), edge_span=5.0)
            carry = rs.integers(2) == 1
            prev_obs = rs.normal(size=obs0.shape).astype(np.float32) if carry else None
            prev_done = rs.integers(0, 7, B).astype(np.uint8) if carry else None
            rst = m.env_reset_pool(tr, int(rs.integers(1 << 40)), int(rs.integers(1 << 20)), int(rs.integers(2)), ego, got[4], ref, virtual, v_light,
                                   cand, cmode, obs0, pool, mask=rmask, obs_src=prev_obs, done_src=prev_done)
            outs.append(got + [masked] + rst)
        names = ['scaled', 'out5', 'd16', 'ego', 'params', 'cand', 'obs', 'done', 'masked obs', 'reset ego', 'reset params', 'reset ref', 'reset virtual',
                 'reset v_light', 'reset done', 'reset cand', 'reset obs']
        for k, (h, d) in enumerate(zip(*outs)):
            if names[k] == 'out5':
                assert np.array_equal(h[0], d[0]) and np.allclose(h[1:], d[1:], rtol=1e-6, atol=0), names[k]
            elif names[k] == 'd16':
                assert np.array_equal(h[:12], d[:12]) and np.allclose(h[12:], d[12:], rtol=1e-6, atol=0), names[k]
            else:
                assert np.array_equal(h, d, equal_nan=True), names[k]
    except AssertionError as e:
        bad += 1
        print('MISMATCH (%s): %s' % (e, tag), flush=True)
    n += 1
print('%d random configurations, %d mismatches' % (n, bad))
