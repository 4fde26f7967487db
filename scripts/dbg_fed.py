#!/usr/bin/env python3
"""Debug aid: repeated fed gated rollouts at 32768 x 32, status after each."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from env_build_amd.dynamics_and_models import EnvironmentModel
from env_build_amd.synthetic import make_rollout_inputs
B, N, H = 16384, 32, 25
dev = torch.device('cuda', 0)
inp = make_rollout_inputs('left', B, N, H, seed=0)
m = EnvironmentModel('left', 0, mode='training', n_veh=N, device=dev)
ego = torch.from_numpy(inp['ego']).to(dev); ref = torch.from_numpy(inp['ref_idx']).to(dev)
trk = m.ref_path.tracking_error_vector_batched(ego[:, 3].contiguous(), ego[:, 4].contiguous(), ego[:, 5].contiguous(), ego[:, 0].contiguous(), 0, ref_indexes=ref).t
obs0 = torch.cat([ego, trk, torch.from_numpy(inp['veh']).to(dev)], 1).contiguous()
tape = torch.from_numpy(inp['actions']).to(dev); live = torch.empty_like(tape)
work, out = torch.empty_like(obs0), torch.empty_like(obs0)
out5 = torch.empty((H, 5, B), device=dev); steps = torch.empty((H,) + tuple(obs0.shape), device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
api, h = m.api, m.handle
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
nb = C.c_int32(); api.rollout_gated_blocks(h, B, C.byref(nb)); nb = nb.value
print('blocks', nb)
sync_each = len(sys.argv) > 1 and sys.argv[1] == 'sync'
R = 12
ready = torch.zeros((R, H), dtype=torch.int32, device=dev); done = torch.zeros((R, H, nb, 16), dtype=torch.int32, device=dev)
status = torch.zeros((R, 2), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
import time
for i in range(R):
    t0 = time.perf_counter()
    api.gate_feed(h, B, H, nb, p(tape), p(live), p(ready[i]), p(done[i]), p(status[i]), 1 << 18, sp, 1, None)
    api.rollout_gated(h, B, H, p(obs0), p(live), p(ref), 0, p(work), p(out), p(out5), p(steps), p(ready[i]), p(done[i]), nb, p(status[i]), 1 << 18, sp)
    if sync_each:
        torch.cuda.synchronize(); print(i, 'ms %.2f' % ((time.perf_counter() - t0) * 1e3), status[i].cpu().tolist(), 'steps done', int(done[i, :, :, 0].all(dim=1).sum()))
torch.cuda.synchronize()
print(status.cpu().tolist(), [int(done[i, :, :, 0].all(dim=1).sum()) for i in range(R)])
