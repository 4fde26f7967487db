#!/usr/bin/env python3
"""Profiling aid: kernel-only timing of eb_rollout_step on cuda:0 (HIP events on the launch stream)."""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from env_build_amd.dynamics_and_models import EnvironmentModel
from env_build_amd.synthetic import make_rollout_inputs

ap = argparse.ArgumentParser()
ap.add_argument('--task', default='left'); ap.add_argument('--n-env', type=int, default=65536)
ap.add_argument('--n-veh', type=int, default=32); ap.add_argument('--mode', default='training')
ap.add_argument('--iters', type=int, default=200); ap.add_argument('--n-future', type=int, default=0, help='look-ahead columns (3 per point): with fp16 storage an odd row width (n_future even) leaves every other row 2 bytes off a dword'); ap.add_argument('--lib', default=None, help='A/B aid: bind this build of libenvbuild_hip.so instead of the in-tree one'); ap.add_argument('--lanes', type=int, default=1, help='independent env sets stepped round-robin (8: the working set leaves the Infinity Cache)'); ap.add_argument('--f16', action='store_true', help='fp16 state storage (configs[4])'); ap.add_argument('--tile', type=int, default=-1, help='eb_debug_set_tile: 0 = 4x8 (2048 records), 1 = 4x4, 2 = 1x4, -1 = by batch size'); ap.add_argument('--scan-prefetch', type=int, default=1, help='eb_debug_set_scan_prefetch: 0 = one group of table entries per loop trip (rounds 1-4), 1 = the first groups in one round trip'); ap.add_argument('--sched', default='-1,-1', help='eb_debug_set_rollout_sched rolling,by_progress: 0 / 1 each, -1 = by grid size'); ap.add_argument('--stage-paths', type=int, default=-1, help='eb_debug_set_stage_paths: the tape / gated kernels keep the path tables in LDS (1) or not (0); -1 = by grid size')
a = ap.parse_args()
dev = torch.device('cuda', 0)
if a.lib:
    from env_build_amd import _capi
    _capi._hip_api = _capi.CApi(a.lib)
inp = make_rollout_inputs(a.task, a.n_env, a.n_veh, 25, seed=0)
m = EnvironmentModel(a.task, a.n_future, mode=a.mode, n_veh=a.n_veh, device=dev)
m.api.debug_set_tile(m.handle, a.tile)
m.api.debug_set_stage_paths(m.handle, a.stage_paths)
m.api.debug_set_scan_prefetch(m.handle, a.scan_prefetch)
m.api.debug_set_rollout_sched(m.handle, *[int(x) for x in a.sched.split(',')])
ego = torch.from_numpy(inp['ego']).to(dev); ref = torch.from_numpy(inp['ref_idx']).to(dev)
if a.mode != 'training': m.ref_path.set_path(1)
trk = m.ref_path.tracking_error_vector_batched(ego[:, 3].contiguous(), ego[:, 4].contiguous(), ego[:, 5].contiguous(),
                                               ego[:, 0].contiguous(), a.n_future, ref_indexes=ref if a.mode == 'training' else None).t
obs0 = torch.cat([ego, trk, torch.from_numpy(inp['veh']).to(dev)], 1).contiguous()
if a.f16: obs0 = obs0.to(torch.float16)
tape = torch.from_numpy(inp['actions']).to(dev)
L = a.lanes
obs0s = [obs0] + [obs0.clone() for _ in range(L - 1)]
bufs = [[torch.empty_like(obs0), torch.empty_like(obs0)] for _ in range(L)]; out5 = [torch.empty((25, 5, a.n_env), device=dev) for _ in range(L)]
p = lambda t: C.c_void_p(t.data_ptr())
st = torch.cuda.current_stream(); sp = C.c_void_p(st.cuda_stream)
fn = m.api.lib.eb_rollout_step_f16 if a.f16 else m.api.lib.eb_rollout_step
def step(i):
    l, t = i % L, (i // L) % 25
    src = obs0s[l] if t == 0 else bufs[l][(t - 1) & 1]
    rc = fn(m.handle, a.n_env, p(src), p(tape[t]), p(ref) if a.mode == 'training' else None, 1, p(bufs[l][t & 1]), p(out5[l][t]), None, sp)
    assert rc == 0, m.api.lib.eb_last_error()
for i in range(50): step(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record(st)
for i in range(a.iters): step(i)
t1 = time.perf_counter()
e1.record(st); torch.cuda.synchronize()
t2 = time.perf_counter()
print('wall: enqueue %.2f us/step, enqueue+drain %.2f us/step; torch events %.2f us/step' % ((t1 - t0) * 1e6 / a.iters, (t2 - t0) * 1e6 / a.iters, e0.elapsed_time(e1) * 1e3 / a.iters))
us = (t2 - t0) * 1e6 / a.iters
alg = ((68 + 16 * a.n_veh + 12 * a.n_future) if a.f16 else (104 + 32 * a.n_veh + 24 * a.n_future)) * a.n_env
print(('f16 ' if a.f16 else '') + 'tile=%d sched=%s lanes=%d ' % (a.tile, a.sched, L) + 'task=%s B=%d N=%d mode=%s: %.2f us/step  %.2f G env-steps/s  alg %.0f GB/s (%.1f%% of 8 TB/s)'
      % (a.task, a.n_env, a.n_veh, a.mode, us, a.n_env / us / 1e3, alg / us / 1e3, alg / us / 1e3 / 80))
