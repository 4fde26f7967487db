#!/usr/bin/env python3
"""Profiling aid: BASELINE configs[1] (4096 x 16, horizon 25) through the three launch forms of the rollout — one launch per
step (eager and hipGraph), the gated one-launch form with every gate open and with eb_gate_feed releasing the steps from a
second stream, and the open-loop tape kernel — us per step and env-steps/s."""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from env_build_amd.dynamics_and_models import EnvironmentModel
from env_build_amd.synthetic import make_rollout_inputs

ap = argparse.ArgumentParser()
ap.add_argument('--n-env', type=int, default=4096); ap.add_argument('--n-veh', type=int, default=16)
ap.add_argument('--horizon', type=int, default=25); ap.add_argument('--reps', type=int, default=200)
a = ap.parse_args()
dev = torch.device('cuda', 0)
B, N, H = a.n_env, a.n_veh, a.horizon
inp = make_rollout_inputs('left', B, N, H, seed=0)
m = EnvironmentModel('left', 0, mode='training', n_veh=N, device=dev)
ego = torch.from_numpy(inp['ego']).to(dev); ref = torch.from_numpy(inp['ref_idx']).to(dev)
trk = m.ref_path.tracking_error_vector_batched(ego[:, 3].contiguous(), ego[:, 4].contiguous(), ego[:, 5].contiguous(),
                                               ego[:, 0].contiguous(), 0, ref_indexes=ref).t
obs0 = torch.cat([ego, trk, torch.from_numpy(inp['veh']).to(dev)], 1).contiguous()
tape = torch.from_numpy(inp['actions']).to(dev)
live = torch.empty_like(tape)
work, out = torch.empty_like(obs0), torch.empty_like(obs0)
out5 = torch.empty((H, 5, B), device=dev)
steps = torch.empty((H,) + tuple(obs0.shape), device=dev)
ready1 = torch.ones(H, dtype=torch.int32, device=dev)
status = torch.zeros(2, dtype=torch.int32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
api, lib, h = m.api, m.api.lib, m.handle
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
nb = C.c_int32(); api.rollout_gated_blocks(h, B, C.byref(nb)); nb = nb.value
done = torch.zeros((H, nb, 16), dtype=torch.int32, device=dev)   # the rollout only ever sets these words: no reset between repetitions
plan = C.c_void_p()
api.plan_create(h, B, H, p(obs0), p(tape), p(ref), 0, p(work), p(out), p(out5), None, None, C.byref(plan))
dst = [out if (H - 1 - t) % 2 == 0 else work for t in range(H)]
src = [obs0] + dst[:-1]
eargs = [(h, B, p(src[t]), p(tape[t]), p(ref), 0, p(dst[t]), p(out5[t]), None, sp) for t in range(H)]

def eager():
    for t in range(H):
        lib.eb_rollout_step(*eargs[t])
def graph():
    lib.eb_plan_launch(plan, sp)
def tape_kernel():
    lib.eb_rollout_tape(h, B, H, p(obs0), p(tape), p(ref), 0, p(work), p(out), p(out5), sp)
def gated_open(publish):
    api.rollout_gated(h, B, H, p(obs0), p(tape), p(ref), 0, p(work), p(out), p(out5), p(steps) if publish else None, p(ready1), p(done),
                      nb, p(status), 1 << 20, sp)
R = a.reps + 5
ready_r = torch.zeros((R, H), dtype=torch.int32, device=dev)      # one set of flags per repetition: nothing to reset in between
done_r = torch.zeros((R, H, nb, 16), dtype=torch.int32, device=dev)
fed_i = [0]
def gated_fed():
    i = fed_i[0] % R
    fed_i[0] += 1
    api.gate_feed(h, B, H, nb, p(tape), p(live), p(ready_r[i]), p(done_r[i]), p(status), 1 << 20, sp, 1, None)
    api.rollout_gated(h, B, H, p(obs0), p(live), p(ref), 0, p(work), p(out), p(out5), p(steps), p(ready_r[i]), p(done_r[i]), nb, p(status), 1 << 20, sp)

def timeit(name, fn, sync_each=False):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
        if sync_each: torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    print('%-58s %8.2f us per %d-step rollout = %6.2f us/step  %6.2f G env-steps/s' % (name, dt * 1e6, H, dt * 1e6 / H, B * H / dt / 1e9))

print('B=%d N=%d H=%d, %d blocks in the gated form' % (B, N, H, nb))
timeit('one launch per step, eager', eager)
timeit('one launch per step, hipGraph replay', graph)
timeit('gated, every gate open, obs published every step', lambda: gated_open(True))
timeit('gated, every gate open, obs not published', lambda: gated_open(False))
timeit('gated, fed step by step by eb_gate_feed (2nd stream)', gated_fed)
timeit('open-loop tape kernel (no gates, nothing published)', tape_kernel)
assert status.cpu().tolist() == [0, 0], status
