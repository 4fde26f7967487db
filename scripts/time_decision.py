#!/usr/bin/env python3
"""Timing aid: one HierarchicalDecision.step (3 path observations -> obj_v -> hysteresis -> 5-step shield -> env step)
for a batch of envs, policy 2 x 256 ELU."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace
from env_build_amd.endtoend_env_utils import VEH_NUM
from env_build_amd.hier_decision import HierarchicalDecision
from env_build_amd.policy import LoadPolicy
ap = argparse.ArgumentParser(); ap.add_argument('--n-env', type=int, default=65536); ap.add_argument('--steps', type=int, default=20)
a = ap.parse_args()
N = VEH_NUM['left']; D = 9 + 4 * N
args = SimpleNamespace(obs_dim=D, act_dim=2, num_hidden_layers=2, num_hidden_units=256, hidden_activation='elu',
                       policy_out_activation='linear', action_range=1.0, deterministic_policy=True, obs_preprocess_type='scale',
                       obs_scale=[0.2] * 6 + [1., 1 / 30., 0.2] + [1 / 30., 1 / 30., 0.2, 1 / 180.] * N)
hd = HierarchicalDecision('left', policy=LoadPolicy(args=args), n_env=a.n_env)
for _ in range(3): hd.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): hd.step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print('n_env=%d: %.1f us per decision step, %.3g env-decisions/s' % (a.n_env, dt * 1e6, a.n_env / dt))
