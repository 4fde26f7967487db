#!/usr/bin/env python3
"""Profiling aid: CrossroadEnd2end.step over a batch of envs (the env-side kernels + the traffic source)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from env_build_amd.endtoend import CrossroadEnd2end
ap = argparse.ArgumentParser()
ap.add_argument('--sizes', default='1,4096,65536'); ap.add_argument('--traffic', default='pool'); ap.add_argument('--steps', type=int, default=50); ap.add_argument('--n-cand', type=int, default=None); ap.add_argument('--separate-flow', action='store_true', help="traffic='flows': eb_traffic_flow_step as a launch of its own (A/B against the flow rule inside the step launch)")
a = ap.parse_args()
for B in [int(x) for x in a.sizes.split(',')]:
    # regimes: (a) auto_reset — the step's own launch resets the envs it finishes (ABI 4; over the flow source: the facade issues
    # the masked reset's launches behind the step's); (b) masked reset after every step
    # (two launches: what a batched driver did before); (c) nobody ever resets (constant random actions drive every ego off the
    # map within a few seconds: the closest-point search then leaves its cell grid for the full scan — the worst case).
    # Each with outputs as arrays of their own (copy_outputs=True, the default) and with the two pre-allocated sets.
    regimes = [('auto reset in the step launch', 'auto'), ('masked reset every step', 'mask'), ('no resets', None)] if B > 1 else [('single env', None)]
    for name, how in regimes:
        for copy in ((True, False) if B > 1 else (True,)):
            env = CrossroadEnd2end('left', n_env=B, multi_display=True, traffic=a.traffic, n_cand=a.n_cand, auto_reset=how == 'auto',
                                   copy_outputs=copy, flow_in_step=not a.separate_flow)
            env.reset()
            act = torch.rand((B, 2), device=env.device) * 2 - 1
            a1 = act[0].cpu().numpy() if B == 1 else act
            def one():
                obs, r, done, info = env.step(a1)
                if how == 'mask': env.reset(mask=done)
            for _ in range(20): one()
            import gc; gc.collect()          # (a collection inside the timed loop once showed up as 100 us per step over 50 steps)
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = a.steps
            for _ in range(n): one()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            print('n_env=%6d traffic=%s, %s, copy_outputs=%s: %.1f us per step, %.3g env-steps/s' % (B, a.traffic, name, copy, dt * 1e6, B / dt))
            del env
