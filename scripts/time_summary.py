#!/usr/bin/env python3
"""Profiling aid: eb_episode_summary through the raw C entry, back to back on one stream (µs per call), at the headline's shard
sizes."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from env_build_amd.dynamics_and_models import EnvironmentModel, _ptr
ap = argparse.ArgumentParser()
ap.add_argument('--sizes', default='4096,32768,65536,262144'); ap.add_argument('--horizon', type=int, default=25); ap.add_argument('--iters', type=int, default=300)
a = ap.parse_args()
for B in [int(x) for x in a.sizes.split(',')]:
    m = EnvironmentModel('left', n_veh=16, device=torch.device('cuda', 0))
    D = m.obs_dim
    out5 = torch.rand((a.horizon, 5, B), device='cuda:0') - 0.3
    final = torch.randn((B, D), device='cuda:0')
    out8 = torch.empty(8, device='cuda:0')
    sp = torch.cuda.current_stream().cuda_stream
    call = lambda: m.api.episode_summary(m.handle, B, a.horizon, _ptr(out5), _ptr(final), _ptr(out8), sp)
    for _ in range(10): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(a.iters): call()
    e1.record(); torch.cuda.synchronize()
    print('n_env=%6d horizon=%d %s: %.2f us per summary   %s' % (B, a.horizon, 'two launches',
          e0.elapsed_time(e1) * 1e3 / a.iters, [round(float(x), 3) for x in out8.tolist()[:6]]), flush=True)
