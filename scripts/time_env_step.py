#!/usr/bin/env python3
"""Profiling aid: CrossroadEnd2end.step over a batch of envs (the env-side kernels + the traffic pool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from env_build_amd.endtoend import CrossroadEnd2end
for B in (1, 4096, 65536):
    env = CrossroadEnd2end('left', n_env=B, multi_display=True)
    env.reset()
    act = torch.rand((B, 2), device=env.device) * 2 - 1
    a1 = act[0].cpu().numpy() if B == 1 else act
    for _ in range(5): env.step(a1)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 50
    for _ in range(n): env.step(a1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print('n_env=%6d: %.1f us per step, %.3g env-steps/s' % (B, dt * 1e6, B / dt))
