#!/usr/bin/env python3
"""Randomised parity sweep (not part of the test suite) of the two ABI-4 forms of eb_env_step: auto_reset (the step that resets
the envs it finished, against step + final rows + eb_env_reset_pool, and against the CPU oracle) and flow (the step that carries
the flow source's rule, against step + eb_traffic_flow_step over a closed loop, and against the CPU oracle) — random task, env
count, candidate / slot counts, future points, tile shape, collision density, light programme.  --waves 4 / 8 forces the waves
per block of the one-launch kernels (eb_debug_set_env_waves): run the sweep a second time with --waves 4 for the four-wave blocks
at these sizes."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tests._helpers import DeviceModel, HostModel, oracle_lib
from tests._env_step_check import auto_reset_case, flow_auto_reset_case, flow_rule_case

ap = argparse.ArgumentParser(); ap.add_argument('--seconds', type=float, default=120.0); ap.add_argument('--seed', type=int, default=0); ap.add_argument('--waves', type=int, default=0)
a = ap.parse_args()
DeviceModel.ENV_WAVES = a.waves
rng = np.random.default_rng(a.seed)
on_gpu, on_cpu = (lambda t, **kw: DeviceModel(t, **kw)), (lambda t, **kw: HostModel(oracle_lib(), t, **kw))
t_end = time.time() + a.seconds
n = {'auto': 0, 'flow': 0, 'flow+auto': 0}; bad = 0
while time.time() < t_end:
    task = ['left', 'straight', 'right'][rng.integers(3)]
    tile = int(rng.choice([-1, 0, 1, 2]))
    seed = int(rng.integers(1 << 30))
    kind = ['auto', 'auto', 'flow', 'flow+auto'][rng.integers(4)]
    try:
        if kind == 'auto':
            NV = [None, 1, 3, 16, 32][rng.integers(5)]
            M = int(rng.choice([1, 2, 7, 16, 23, 40, 60, 64]))
            B = int(rng.choice([1, 15, 16, 17, 63, 64, 65, 300, 1500]))
            nf = int(rng.choice([0, 0, 2]))
            cp = float(rng.choice([0.0, 0.02, 0.3]))
            vn = bool(rng.integers(4) == 0)
            tag = 'auto %s NV=%s M=%d B=%d nf=%d tile=%d close=%.2f v_light_none=%d seed=%d' % (task, NV, M, B, nf, tile, cp, vn, seed)
            g = auto_reset_case(on_gpu, task, B, M, NV, nf, tile, cp, seed, vn, strict=False)
            w = auto_reset_case(on_cpu, task, B, M, NV, nf, None, cp, seed, vn, strict=False)
            for k, (x, y) in enumerate(zip(g, w)):
                if y is None:
                    assert x is None, k
                elif k in (1, 2):
                    assert np.allclose(x, y, rtol=1e-6, atol=0), k
                else:
                    assert np.array_equal(x, y, equal_nan=True), k
        elif kind == 'flow+auto':   # ABI 5: the flow source's own reset in the step kernel's tail (one lane per (finished env, slot))
            K = int(rng.choice([1, 2, 3, 5]))
            B = int(rng.choice([1, 16, 17, 65, 300]))
            steps = int(rng.choice([3, 8, 16]))
            tag = 'flow+auto %s K=%d B=%d steps=%d tile=%d seed=%d' % (task, K, B, steps, tile, seed)
            g = flow_auto_reset_case(on_gpu, task, B, K, steps, tile, seed, strict=False)
            w = flow_auto_reset_case(on_cpu, task, B, K, steps, None, seed, strict=False)
            for t, (gs, ws) in enumerate(zip(g, w)):
                for k, (x, y) in enumerate(zip(gs, ws)):
                    if y is None:
                        assert x is None, (t, k)
                    elif x.dtype == np.float32 and x.ndim == 2 and x.shape[0] in (5, 16) and x.shape[1] == B:
                        assert np.allclose(x, y, rtol=1e-6, atol=0), (t, k)
                    else:
                        assert np.array_equal(x, y, equal_nan=True), (t, k)
        else:
            K = int(rng.choice([1, 2, 3, 5]))
            B = int(rng.choice([1, 16, 17, 65, 300]))
            steps = int(rng.choice([3, 12, 40]))
            lc = int(rng.choice([0, 1, 7]))
            tag = 'flow %s K=%d B=%d steps=%d tile=%d light_cycle=%d seed=%d' % (task, K, B, steps, tile, lc, seed)
            g = flow_rule_case(on_gpu, task, B, K, steps, tile, lc, seed, strict=False)
            w = flow_rule_case(on_cpu, task, B, K, steps, None, lc, seed, strict=False)
            for t, (gs, ws) in enumerate(zip(g, w)):
                for k, (x, y) in enumerate(zip(gs, ws)):
                    if k in (1, 2):
                        assert np.allclose(x, y, rtol=1e-6, atol=0), (t, k)
                    else:
                        assert np.array_equal(x, y, equal_nan=True), (t, k)
    except AssertionError as e:
        bad += 1
        print('MISMATCH (%s): %s' % (e, tag), flush=True)
    n[kind] += 1
print('%d auto-reset, %d flow-rule and %d flow-rule + auto-reset configurations, %d mismatches' % (n['auto'], n['flow'], n['flow+auto'], bad))
