#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched model rollout (EnvironmentModel.rollout_out) on MI355X.

A "step" is one rollout_out call over the whole batch: B env-steps.  Workload at N GPUs:
BASELINE.json configs[2] per GPU (N_env = 65 536, N_veh = 32, horizon 25, task `left`, training
mode, fp32), i.e. weak scaling — every rank owns an independent shard of envs, there is no
data-path collective, and the only exchange is one all-gather (RCCL) of the per-rank episodic-return
summary at the end of every 25-step horizon (inside the timed region).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     — algorithmic bytes per launch (104 + 32*N_veh per env-step) divided
                 by the rollout kernel's average launch duration measured with HIP events on the
                 launch stream, against the 8 TB/s HBM peak;
  cpu_baseline — the CPU oracle (oracle/, plain C port of the reference path, OpenMP over envs) timed
                 on this box's host cores on a bounded sample of the same workload.  The oracle is
                 only the thing timed here, never part of the GPU path.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TASK, N_ENV, N_VEH, HORIZON = 'left', 65536, 32, 25
ALG_BYTES_PER_ENV_STEP = 104 + 32 * N_VEH        # fp32, SURVEY.md §8(d): 1128 B at N_veh = 32
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(inp, obs0, budget_s=12.0):
    """Oracle timed on host cores: bounded sample (B_cpu envs x HORIZON steps, repeated until the
    budget is used), all cores via OpenMP and then 1 core (the reference pins TF to 1 thread)."""
    from tests._helpers import HostModel, oracle_lib
    api = oracle_lib()
    api.lib.eb_oracle_set_threads.restype = C.c_int
    api.lib.eb_oracle_set_threads.argtypes = [C.c_int]
    b_cpu = 8192
    host = HostModel(api, TASK, n_veh=N_VEH)
    obs, act, ref = obs0[:b_cpu].copy(), inp['actions'][:, :b_cpu].copy(), inp['ref_idx'][:b_cpu].copy()
    res = {}
    for label, threads in (('all', os.cpu_count() or 1), ('one', 1)):
        used = api.lib.eb_oracle_set_threads(int(threads))
        host.rollout_tape(obs, act[:2], ref)   # warm-up
        n_steps, t0 = 0, time.perf_counter()
        while True:
            host.rollout_tape(obs, act, ref)
            n_steps += HORIZON
            if time.perf_counter() - t0 > budget_s / 2:
                break
        dt = time.perf_counter() - t0
        res[label] = (b_cpu * n_steps / dt, used, n_steps)
    v_all, cores, n_steps = res['all']
    return {'value': v_all, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
            'value_1core': res['one'][0],
            'sample': '%d envs x %d steps (N_veh=%d, same seeded inputs), oracle/envbuild_oracle.c, '
                      'OpenMP over envs; 1-core figure is the reference-faithful setting (TF pinned to 1 thread)'
                      % (b_cpu, n_steps, N_VEH)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from env_build_amd import _capi
    from env_build_amd.dynamics_and_models import EnvironmentModel
    from env_build_amd.synthetic import make_rollout_inputs, assemble_obs

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=dev)

    # ---- synthetic shard of this rank (seed = rank: independent envs per GPU) ----
    inp = make_rollout_inputs(TASK, N_ENV, N_VEH, HORIZON, seed=rank)
    model = EnvironmentModel(TASK, num_future_data=0, mode='training', n_veh=N_VEH, device=dev)
    ego = torch.from_numpy(inp['ego']).to(dev)
    trk = model.ref_path.tracking_error_vector_batched(ego[:, 3].contiguous(), ego[:, 4].contiguous(),
                                                       ego[:, 5].contiguous(), ego[:, 0].contiguous(), 0,
                                                       ref_indexes=torch.from_numpy(inp['ref_idx']).to(dev)).t
    obs0 = torch.cat([ego, trk, torch.from_numpy(inp['veh']).to(dev)], 1).contiguous()
    ref_idx = torch.from_numpy(inp['ref_idx']).to(dev)
    tape = torch.from_numpy(inp['actions']).to(dev)                      # [H, B, 2]
    bufs = [torch.empty_like(obs0), torch.empty_like(obs0)]
    out5 = torch.empty((HORIZON, 5, N_ENV), dtype=torch.float32, device=dev)
    summary_all = torch.zeros((world, 6), dtype=torch.float32, device=dev)

    api, h = model.api, model.handle
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    step_fn = api.lib.eb_rollout_step
    tape_p = [p(tape[t]) for t in range(HORIZON)]
    out5_p = [p(out5[t]) for t in range(HORIZON)]
    obs0_p, buf_p, ref_p = p(obs0), [p(bufs[0]), p(bufs[1])], p(ref_idx)

    def one_step(i):
        """step i of the job: horizon-periodic, reading obs0 at the start of each horizon."""
        t = i % HORIZON
        src = obs0_p if t == 0 else buf_p[(t - 1) & 1]
        rc = step_fn(h, N_ENV, src, tape_p[t], ref_p, 0, buf_p[t & 1], out5_p[t], None, sp)
        if rc != 0:
            api.check(rc)
        if t == HORIZON - 1:
            # episodic-return summary of this shard: sum reward, sum punish_train, sum punish_real,
            # #envs with a real collision/road penalty, mean |delta_y|, max |delta_y| of the final obs
            s = out5.sum(dim=(0, 2))
            fin = bufs[t & 1][:, 6].abs()
            mine = torch.stack([s[0], s[1], s[2], (out5[:, 2] > 0).any(0).sum().float(), fin.mean(), fin.max()])
            if world > 1:
                dist.all_gather_into_tensor(summary_all.view(-1), mine)
            else:
                summary_all[0].copy_(mine)

    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        one_step(i)
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())

    if rank == 0:
        value = N_ENV * world * args.steps / dt_max
        launch_s = ev_ms * 1e-3 / args.steps       # HIP events on the launch stream, per rollout launch
        achieved = ALG_BYTES_PER_ENV_STEP * N_ENV / launch_s / 1e9
        line = {
            'metric': 'env-steps/s (batched rollout) at N_env x N_veh; achieved HBM GB/s vs peak',
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt_max * 1e3 / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'configs[2]: N_env=65536 per GPU, N_veh=32, horizon=25, task=left, '
                                   'mode=training, closed-loop rollout_out (one launch per step)',
                       'n_env_per_gpu': N_ENV, 'n_veh': N_VEH, 'horizon': HORIZON,
                       'parallelism': 'env-shard x%d, all-gather of the episodic summary per horizon' % world},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                         'kernel': 'eb::rollout_step_kernel<0>', 'alg_bytes_per_launch': ALG_BYTES_PER_ENV_STEP * N_ENV,
                         'avg_launch_us': launch_s * 1e6},
            'summary': [float(x) for x in summary_all[0].tolist()],
        }
        if not args.no_cpu_baseline:
            obs0_h = obs0.cpu().numpy()
            line['cpu_baseline'] = cpu_baseline(inp, obs0_h)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
