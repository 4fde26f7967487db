#!/usr/bin/env python3
"""Experiment: the HBM-resident rollout (8 independent env sets of 65 536 x 32, round-robin) on 1 / 2 / 3 / 4 / 8 HIP streams."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from env_build_amd.dynamics_and_models import EnvironmentModel
dev = torch.device('cuda', 0)
m = EnvironmentModel(bench.TASK, 0, mode='training', n_veh=32, device=dev)
for lanes, n_env in ((2, 32768), (4, 16384), (3, 21846), (2, 65536), (4, 65536)):
    for st in (1, 2, 3, 4):
        if st > lanes: continue
        r = bench.side_config(torch, None, m, n_env, 32, 0, 100, 25, 5, lanes=lanes, forms=('eager',), streams=st)
        print('lanes %d x %6d envs, %d streams: %.2f us per launch, %.1f %% of the HBM peak, %.2f G env-steps/s' % (lanes, n_env, st, r['avg_launch_us'] if st == 1 else r['ms_per_step'] * 1e3, 100 * r['frac'], r['value'] / 1e9), flush=True)
