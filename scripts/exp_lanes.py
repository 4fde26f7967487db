#!/usr/bin/env python3
"""Experiment: the 65 536-env rollout as L independent lanes (contiguous env ranges, own hipGraph plan, own
stream) replayed concurrently — do the launch / drain gaps of one lane fill with the other lane's kernels?"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from env_build_amd.dynamics_and_models import EnvironmentModel
from env_build_amd.synthetic import make_rollout_inputs

ap = argparse.ArgumentParser()
ap.add_argument('--n-env', type=int, default=65536); ap.add_argument('--n-veh', type=int, default=32)
ap.add_argument('--lanes', type=int, default=2); ap.add_argument('--reps', type=int, default=40)
a = ap.parse_args()
dev = torch.device('cuda', 0)
B, N, H, L = a.n_env, a.n_veh, 25, a.lanes
inp = make_rollout_inputs('left', B, N, H, seed=0)
m = EnvironmentModel('left', 0, mode='training', n_veh=N, device=dev)
ego = torch.from_numpy(inp['ego']).to(dev); ref = torch.from_numpy(inp['ref_idx']).to(dev)
trk = m.ref_path.tracking_error_vector_batched(ego[:, 3].contiguous(), ego[:, 4].contiguous(), ego[:, 5].contiguous(),
                                               ego[:, 0].contiguous(), 0, ref_indexes=ref).t
obs0 = torch.cat([ego, trk, torch.from_numpy(inp['veh']).to(dev)], 1).contiguous()
tape = torch.from_numpy(inp['actions']).to(dev)
p = lambda t: C.c_void_p(t.data_ptr())
api, h = m.api, m.handle
per = B // L
lanes = []
for l in range(L):
    lo, hi = l * per, (l + 1) * per
    o = obs0[lo:hi].contiguous(); tp = tape[:, lo:hi].contiguous(); r = ref[lo:hi].contiguous()
    w, f = torch.empty_like(o), torch.empty_like(o); o5 = torch.empty((H, 5, per), device=dev)
    plan = C.c_void_p()
    api.plan_create(h, per, H, p(o), p(tp), p(r), 0, p(w), p(f), p(o5), None, None, C.byref(plan))
    lanes.append(dict(plan=plan, stream=torch.cuda.Stream(device=dev), keep=(o, tp, r, w, f, o5)))
def run():
    for ln in lanes:
        api.plan_launch(ln['plan'], C.c_void_p(ln['stream'].cuda_stream))
for _ in range(5): run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps): run()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
us = dt / (a.reps * H) * 1e6
print('lanes %d: %.2f us per step of %d envs  (%.2f G env-steps/s, %.1f%% of 8 TB/s)' % (L, us, B, B / us / 1e3, (104 + 32 * N) * B / (us * 1e-6) / 8e12 * 100))
