#!/usr/bin/env python3
"""Tuning aid: env-step time at one batch size.   usage: sweep_env_tile.py <n_env> [tile -1|0|1|2 = auto|64|32|16 envs] [waves 0|4|8]
(eb_debug_set_tile / eb_debug_set_env_waves on the env's handle; the thresholds in env_step_tile_envs come from this sweep)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
n = int(sys.argv[1])
tile = int(sys.argv[2]) if len(sys.argv) > 2 else -1
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 0
r = bench.env_step_bench(torch, torch.device('cuda', 0), n, tile=tile, waves=waves)
print('n_env=%d tile=%s waves=%s: %.2f us' % (n, {-1: 'auto', 0: 64, 1: 32, 2: 16}[tile], waves or 'auto', r['avg_launch_us']))
