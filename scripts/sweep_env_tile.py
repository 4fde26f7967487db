#!/usr/bin/env python3
"""Tuning aid: env-step time by batch size and tile shape (EB_ENV_TILE is read once per process: one size x tile per run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
n = int(sys.argv[1])
r = bench.env_step_bench(torch, torch.device('cuda', 0), n)
print('n_env=%d tile=%s: %.2f us' % (n, os.environ.get('EB_ENV_TILE', 'auto'), r['avg_launch_us']))
