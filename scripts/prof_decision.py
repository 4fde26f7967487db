#!/usr/bin/env python3
"""Timing aid: where a HierarchicalDecision.step goes (synchronising after every stage)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from env_build_amd.endtoend_env_utils import VEH_NUM
from env_build_amd.hier_decision import HierarchicalDecision
from env_build_amd.policy import LoadPolicy
from env_build_amd.dynamics_and_models import _unwrap
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = VEH_NUM['left']; D = 9 + 4 * N
args = SimpleNamespace(obs_dim=D, act_dim=2, num_hidden_layers=2, num_hidden_units=256, hidden_activation='elu',
                       policy_out_activation='linear', action_range=1.0, deterministic_policy=True, obs_preprocess_type='scale',
                       obs_scale=[0.2] * 6 + [1., 1 / 30., 0.2] + [1 / 30., 1 / 30., 0.2, 1 / 180.] * N)
hd = HierarchicalDecision('left', policy=LoadPolicy(args=args), n_env=B, auto_reset=False)
for _ in range(3): hd.step()
T = {}
def tick(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return r
n = 10
for _ in range(n):
    all_obs = tick('path_observations (3 x get_obs)', hd.path_observations)
    pv = tick('obj_value_batch [3B rows]', lambda: _unwrap(hd.policy.obj_value_batch(all_obs.reshape(-1, all_obs.shape[2]))).reshape(-1, B))
    idx = tick('select_path', lambda: hd.select_path(pv))
    obs_real = tick('gather obs', lambda: all_obs.gather(0, idx.view(1, B, 1).expand(1, B, all_obs.shape[2]))[0].contiguous())
    act = tick('safe_shield (5-step shield + run_batch)', lambda: hd.safe_shield(obs_real, idx))
    hd.env._ref_idx.copy_(idx.to(torch.int32))
    out = tick('env.step', lambda: hd.env.step(act[0]))
    d = _unwrap(out[2]).bool()
    tick('reset(mask)', lambda: hd.reset(mask=d))
for k, v in T.items():
    print('%-45s %8.1f us' % (k, v / n * 1e6))
print('%-45s %8.1f us' % ('sum', sum(T.values()) / n * 1e6))
