#!/usr/bin/env python3
"""PCIe-inclusive rate of the Python facade when the caller hands over HOST arrays: every
rollout_out copies obs + actions to the GPU and the six results back.  Never used as bench.py's `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from env_build_amd.dynamics_and_models import EnvironmentModel
from env_build_amd.synthetic import assemble_obs, make_rollout_inputs

B, N, H = 65536, 32, 25
inp = make_rollout_inputs('left', B, N, H, seed=0)
m = EnvironmentModel('left', 0, mode='training', n_veh=N)
trk = m.ref_path.tracking_error_vector_batched(inp['ego'][:, 3], inp['ego'][:, 4], inp['ego'][:, 5], inp['ego'][:, 0], 0,
                                               ref_indexes=inp['ref_idx']).numpy()
obs = assemble_obs(inp['ego'], trk, inp['veh'])
def run(n):
    o = obs
    for t in range(n):
        m.reset(o, inp['ref_idx'])                       # host -> device
        out = m.rollout_out(inp['actions'][t % H])        # host actions -> device, one kernel
        o = out[0].numpy()                                # device -> host
        _ = [x.numpy() for x in out[1:]]
    torch.cuda.synchronize()
run(3)
t0 = time.perf_counter(); n = 20; run(n); dt = time.perf_counter() - t0
print('host-array round trip: %.2f ms per step, %.1f M env-steps/s (PCIe + pageable-memory copies included; obs %.1f MB each way)'
      % (dt / n * 1e3, B * n / dt / 1e6, obs.nbytes / 1e6))
