#!/usr/bin/env python3
"""Timing aid for §8(f) rank 2: the fused MLP kernel (TFLOP/s against the 157 TFLOP/s f32 matrix peak) and the
5-step / 20-step shield at configs[2] size.  HIP events on the launch stream, median of repeats."""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from types import SimpleNamespace
from env_build_amd.dynamics_and_models import EnvironmentModel
from env_build_amd.policy import LoadPolicy
from env_build_amd.shield import is_safe
from env_build_amd.synthetic import make_rollout_inputs

ap = argparse.ArgumentParser()
ap.add_argument('--n-env', type=int, default=65536); ap.add_argument('--n-veh', type=int, default=32)
ap.add_argument('--units', type=int, default=256); ap.add_argument('--hidden', type=int, default=2)
ap.add_argument('--lib', default=None, help='A/B aid: bind this build of libenvbuild_hip.so instead of the in-tree one'); ap.add_argument('--reps', type=int, default=20); ap.add_argument('--act', default='elu'); ap.add_argument('--no-shield', action='store_true')
a = ap.parse_args()
dev = torch.device('cuda', 0)
if a.lib:
    from env_build_amd import _capi
    _capi._hip_api = _capi.CApi(a.lib)
B, N = a.n_env, a.n_veh
model = EnvironmentModel('left', 0, mode='training', n_veh=N, device=dev)
D = model.obs_dim
args = SimpleNamespace(obs_dim=D, act_dim=2, num_hidden_layers=a.hidden, num_hidden_units=a.units, hidden_activation=a.act,
                       policy_out_activation='linear', action_range=1.0, deterministic_policy=True, obs_preprocess_type='scale',
                       obs_scale=[0.2] * 6 + [1., 1 / 30., 0.2] + [1 / 30., 1 / 30., 0.2, 1 / 180.] * N)
pol = LoadPolicy(args=args, device=dev)
inp = make_rollout_inputs('left', B, N, 25, seed=0)
ego = torch.from_numpy(inp['ego']).to(dev); ref = torch.from_numpy(inp['ref_idx']).to(dev)
trk = model.ref_path.tracking_error_vector_batched(ego[:, 3].contiguous(), ego[:, 4].contiguous(), ego[:, 5].contiguous(),
                                                   ego[:, 0].contiguous(), 0, ref_indexes=ref).t
obs0 = torch.cat([ego, trk, torch.from_numpy(inp['veh']).to(dev)], 1).contiguous()

def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts)), float(np.min(ts))

net = pol.policy.policy
out = torch.empty((B, 2), dtype=torch.float32, device=dev)
lib = net.api.lib
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run_policy():
    assert lib.eb_policy_run_batch(net._handle, B, C.c_void_p(obs0.data_ptr()), C.c_float(1.0), C.c_void_p(out.data_ptr()), sp) == 0
flops = 2.0 * B * (D * a.units + (a.hidden - 1) * a.units * a.units + a.units * 4)
med, mn = timed(run_policy, a.reps)
res = {'n_env': B, 'obs_dim': D, 'net': '%d -> %s -> 4 (%s)' % (D, ' -> '.join([str(a.units)] * a.hidden), a.act),
       'policy_us': med, 'policy_us_min': mn, 'algorithmic_gflop': flops / 1e9, 'tflops': flops / med / 1e6,
       'frac_of_157_tflops': flops / med / 1e6 / 157.3}
model.reset(obs0, ref)
for steps, pen in (() if a.no_shield else ((5, 'veh2veh4real'), (20, 'real_punish_term'))):
    med, mn = timed(lambda: is_safe(model, pol, obs0, steps=steps, penalty=pen), max(5, a.reps // 2))
    res['shield_%d_steps_us' % steps] = med
    res['shield_%d_steps_states_per_s' % steps] = B / med * 1e6
print(json.dumps(res))
