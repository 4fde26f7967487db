#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched model rollout (EnvironmentModel.rollout_out) on MI355X.

A "step" is one rollout_out pass over the whole batch = B env-steps.  Workload at N GPUs:
BASELINE.json configs[2] per GPU (N_env = 65 536, N_veh = 32, horizon 25, task `left`, training
mode, fp32) — weak scaling: every rank owns an independent shard of envs, there is no data-path
collective; the only exchange is one all-gather (RCCL) of the 8-float episodic-return summary at the
end of every 25-step horizon, inside the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The K timed steps run as K // 25 rollouts of 25 launches (one kernel launch per rollout step — the
policy-in-the-loop form, not a fused open-loop kernel), either as replays of a 25-launch hipGraph (eb_plan_*,
--graph) or as 25 eager eb_rollout_step calls (--eager); by default the warm-up times both and the timed
region uses the faster.  Rank 0 prints ONE JSON line with two extra objects:
  roofline     — algorithmic bytes per launch (104 + 32*N_veh per env-step) divided
                 by the rollout kernel's average launch duration, measured with HIP event pairs
                 (eb_event_*) on the launch stream around every graph replay — so the figure includes
                 the inter-kernel gaps — against the 8 TB/s HBM peak;
  cpu_baseline — the CPU oracle (oracle/, plain-C port of the reference path, OpenMP over envs) timed
                 on this box's host cores on a bounded sample of the same workload.  The oracle is
                 only the thing timed there, never part of the GPU path.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import env_build_amd  # noqa: E402,F401  (sets HIP_FORCE_DEV_KERNARG before the HIP runtime starts)

TASK, N_ENV, N_VEH, HORIZON = 'left', 65536, 32, 25
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8 TB/s spec
MAX_EVENT_PAIRS = 256


def alg_bytes_per_env_step(n_veh):
    return 104 + 32 * n_veh                      # fp32, SURVEY.md §8(d): 1128 B at N_veh = 32


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(inp, obs0, n_veh, budget_s=18.0):
    """Oracle timed on host cores on a bounded sample (8192 envs x HORIZON steps, repeated until the budget is
    used): 1 thread (the reference pins TF to 1 thread) and OpenMP over envs on 8 / 32 / all usable cores
    (a cgroup CPU quota can make "all" slower than 8, so every count is reported and the best one is `value`)."""
    from tests._helpers import HostModel, oracle_lib
    api = oracle_lib()
    api.lib.eb_oracle_set_threads.restype = C.c_int
    api.lib.eb_oracle_set_threads.argtypes = [C.c_int]
    b_cpu = 8192
    host = HostModel(api, TASK, n_veh=n_veh)
    obs, act, ref = obs0[:b_cpu].copy(), inp['actions'][:, :b_cpu].copy(), inp['ref_idx'][:b_cpu].copy()
    counts = sorted(set([1] + [c for c in (8, 32) if c < host_cores()] + [host_cores()]))
    res = {}
    for threads in counts:
        used = api.lib.eb_oracle_set_threads(int(threads))
        host.rollout_tape(obs, act[:2], ref)   # warm-up
        n_steps, t0 = 0, time.perf_counter()
        while True:
            host.rollout_tape(obs, act, ref)
            n_steps += HORIZON
            if time.perf_counter() - t0 > budget_s / len(counts):
                break
        dt = time.perf_counter() - t0
        res[used] = (b_cpu * n_steps / dt, n_steps)
    best = max(res, key=lambda k: res[k][0])
    return {'value': res[best][0], 'unit': 'env-steps/s', 'cores': best, 'kind': 'port',
            'value_1core': res[1][0], 'by_threads': {str(k): v[0] for k, v in res.items()},
            'sample': '%d envs x %s steps on %s threads (N_veh=%d, same seeded inputs), oracle/envbuild_oracle.c, OpenMP '
                      'over envs; the 1-thread figure is the reference-faithful setting (the reference pins TF to 1 thread)'
                      % (b_cpu, '/'.join(str(v[1]) for v in res.values()), '/'.join(str(k) for k in res), n_veh)}


def shield_bench(args):
    """The model-predictive safety shield with the policy network on the GPU (hier_decision.py:89-97): one JSON line in
    the same format; the roofline object is the policy kernel's (f32 matrix cores), the bound of this loop."""
    import torch
    from types import SimpleNamespace
    from env_build_amd.dynamics_and_models import EnvironmentModel
    from env_build_amd.policy import LoadPolicy
    from env_build_amd.synthetic import make_rollout_inputs
    if args.gpus != 1:
        raise SystemExit('--shield runs on one GPU')
    B, N, steps_ahead, units, hidden = args.n_env, args.n_veh, 5, 256, 2
    dev = torch.device('cuda', 0)
    model = EnvironmentModel(TASK, 0, mode='training', n_veh=N, device=dev)
    D = model.obs_dim
    pargs = SimpleNamespace(obs_dim=D, act_dim=2, num_hidden_layers=hidden, num_hidden_units=units, hidden_activation='elu',
                            policy_out_activation='linear', action_range=1.0, deterministic_policy=True,
                            obs_preprocess_type='scale',
                            obs_scale=[0.2] * 6 + [1., 1 / 30., 0.2] + [1 / 30., 1 / 30., 0.2, 1 / 180.] * N)
    pol = LoadPolicy(args=pargs, device=dev)                      # random orthogonal weights (no checkpoints travel)
    inp = make_rollout_inputs(TASK, B, N, HORIZON, seed=0)
    ego = torch.from_numpy(inp['ego']).to(dev)
    ref = torch.from_numpy(inp['ref_idx']).to(dev)
    trk = model.ref_path.tracking_error_vector_batched(ego[:, 3].contiguous(), ego[:, 4].contiguous(), ego[:, 5].contiguous(),
                                                       ego[:, 0].contiguous(), 0, ref_indexes=ref).t
    obs0 = torch.cat([ego, trk, torch.from_numpy(inp['veh']).to(dev)], 1).contiguous()
    net = pol.policy.policy
    api, lib = model.api, model.api.lib
    p = lambda t: C.c_void_p(t.data_ptr())
    a, b = torch.empty_like(obs0), torch.empty_like(obs0)
    act = torch.empty((B, 2), dtype=torch.float32, device=dev)
    out5 = torch.empty((5, B), dtype=torch.float32, device=dev)
    punish = torch.empty((B,), dtype=torch.float32, device=dev)
    safe = torch.empty((B,), dtype=torch.uint8, device=dev)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def shield():
        rc = lib.eb_shield_is_safe(model.handle, net._handle, B, p(obs0), p(ref), 0, steps_ahead, 0, C.c_float(1.0), p(a), p(b),
                                   p(act), p(out5), p(punish), p(safe), sp)
        if rc != 0:
            api.check(rc)

    def policy_only():
        rc = lib.eb_policy_run_batch(net._handle, B, p(obs0), C.c_float(1.0), p(act), sp)
        if rc != 0:
            api.check(rc)

    for _ in range(max(1, args.warmup // 10)):
        shield()
    ev = []
    for _ in range(2):
        e = C.c_void_p()
        api.event_create(model.handle, C.byref(e))
        ev.append(e)
    lib.eb_event_record(ev[0], sp)
    for _ in range(20):
        policy_only()
    lib.eb_event_record(ev[1], sp)
    ms = C.c_float()
    api.event_elapsed_ms(ev[0], ev[1], C.byref(ms))
    policy_s = ms.value * 1e-3 / 20
    n_pass = max(1, args.steps // 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_pass):
        shield()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flops = 2.0 * B * (D * units + (hidden - 1) * units * units + units * 4)
    line = {
        'metric': 'shield checks/s (start states through a %d-step policy-in-the-loop look-ahead)' % steps_ahead,
        'value': B * n_pass / dt, 'unit': 'states/s', 'n_gpus': 1, 'steps': n_pass, 'warmup': max(1, args.warmup // 10),
        'ms_per_step': dt * 1e3 / n_pass, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': 'SEPARATE figure: eb_shield_is_safe on configs[2] states (N_env=%d, N_veh=%d, task=%s): %d x [MLPNet '
                               '%d -> %s -> 4 (ELU, random orthogonal weights) -> rollout_out], penalty veh2veh4real'
                               % (B, N, TASK, steps_ahead, D, ' -> '.join([str(units)] * hidden)),
                   'n_env_per_gpu': B, 'n_veh': N, 'look_ahead': steps_ahead, 'parallelism': 'single GPU'},
        'roofline': {'bound': 'mfma', 'achieved': flops / policy_s / 1e12, 'peak': 157.3, 'unit': 'TFLOP/s',
                     'frac': flops / policy_s / 1e12 / 157.3, 'traffic': None, 'kernel': 'eb::mlp_kernel<2, 2>',
                     'alg_flop_per_launch': flops, 'avg_launch_us': policy_s * 1e6, 'launches_timed': 20},
        'cpu_baseline': None,
        'unsafe_fraction': float(1.0 - safe.float().mean().item()),
    }
    print(json.dumps(line))
    for e in ev:
        api.event_destroy(e)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--n-env', type=int, default=N_ENV, help='envs per GPU (default: configs[2])')
    ap.add_argument('--n-veh', type=int, default=N_VEH)
    ap.add_argument('--eager', action='store_true', help='one host launch per step instead of hipGraph replays')
    ap.add_argument('--graph', action='store_true', help='hipGraph replays (default: whichever of the two the warm-up finds faster)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--open-loop', action='store_true',
                    help='SEPARATE figure (SURVEY.md §8(f)1): eb_rollout_tape, the whole 25-step tape in one launch with the '
                         'state in registers — VALU-bound, not the HBM-bound closed-loop headline')
    ap.add_argument('--shield', action='store_true',
                    help='SEPARATE figure (SURVEY.md §8(f)2): eb_shield_is_safe — 5 x [policy MLP (137 -> 256 -> 256 -> 4, ELU) -> '
                         'rollout step] per start state; a "step" is one shield pass over the batch; N = 1 only')
    args = ap.parse_args()
    n_env, n_veh = args.n_env, args.n_veh
    if args.shield:
        return shield_bench(args)

    import torch
    import torch.distributed as dist
    from env_build_amd.dynamics_and_models import EnvironmentModel
    from env_build_amd.synthetic import make_rollout_inputs
    from env_build_amd.sharding import combine_summaries, gather_summaries_async

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or os.environ.get('EB_BENCH_FORCE_DIST') == '1'   # the latter: exercise the RCCL calls on one GPU
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=dev)

    # ---- synthetic shard of this rank (seed = rank: independent envs per GPU), resident in HBM ----
    inp = make_rollout_inputs(TASK, n_env, n_veh, HORIZON, seed=rank)
    model = EnvironmentModel(TASK, num_future_data=0, mode='training', n_veh=n_veh, device=dev)
    ego = torch.from_numpy(inp['ego']).to(dev)
    ref_idx = torch.from_numpy(inp['ref_idx']).to(dev)
    trk = model.ref_path.tracking_error_vector_batched(ego[:, 3].contiguous(), ego[:, 4].contiguous(),
                                                       ego[:, 5].contiguous(), ego[:, 0].contiguous(), 0,
                                                       ref_indexes=ref_idx).t
    obs0 = torch.cat([ego, trk, torch.from_numpy(inp['veh']).to(dev)], 1).contiguous()
    tape = torch.from_numpy(inp['actions']).to(dev)                      # [H, B, 2]
    work, final = torch.empty_like(obs0), torch.empty_like(obs0)
    out5 = torch.empty((HORIZON, 5, n_env), dtype=torch.float32, device=dev)
    # two summary buffers: the all-gather of rollout k overlaps the kernels of rollout k+1 (own RCCL stream)
    summaries = [torch.zeros((8,), dtype=torch.float32, device=dev) for _ in range(2)]
    in_flight = [None, None]
    gathered = [torch.zeros((world, 8), dtype=torch.float32, device=dev)]
    n_rollouts = [0]

    api, h = model.api, model.handle
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    plan = C.c_void_p()
    api.plan_create(h, n_env, HORIZON, p(obs0), p(tape), p(ref_idx), 0, p(work), p(final), p(out5), None, C.byref(plan))
    lib = api.lib
    # the eager form ping-pongs exactly as eb_rollout_tape does, so that step 24 lands in `final`
    dst = [final if (HORIZON - 1 - t) % 2 == 0 else work for t in range(HORIZON)]
    src = [obs0] + dst[:-1]
    eager_args = [(h, n_env, p(src[t]), p(tape[t]), p(ref_idx), 0, p(dst[t]), p(out5[t]), None, sp) for t in range(HORIZON)]

    def tape_launch():
        rc = lib.eb_rollout_tape(h, n_env, HORIZON, p(obs0), p(tape), p(ref_idx), 0, p(work), p(final), p(out5), sp)
        if rc != 0:
            api.check(rc)

    def eager_steps(t0, t1):
        for t in range(t0, t1):
            rc = lib.eb_rollout_step(*eager_args[t])
            if rc != 0:
                api.check(rc)

    def end_of_horizon():
        # episodic-return summary of this shard (two small kernels), then the only inter-GPU exchange
        slot = n_rollouts[0] & 1
        n_rollouts[0] += 1
        if in_flight[slot] is not None:
            gathered[0] = in_flight[slot].result()   # the gather of two rollouts ago: long finished
        api.episode_summary(h, n_env, HORIZON, p(out5), p(final), p(summaries[slot]), sp)
        in_flight[slot] = gather_summaries_async(summaries[slot])   # env_build_amd/sharding.py: 8 floats per rank

    def drain_gathers():
        order = [n_rollouts[0] & 1, (n_rollouts[0] + 1) & 1]       # oldest first
        for slot in order:
            if in_flight[slot] is not None:
                gathered[0] = in_flight[slot].result()
                in_flight[slot] = None

    n_pairs = min(MAX_EVENT_PAIRS, max(1, args.steps // HORIZON))
    ev = []
    for _ in range(2 * n_pairs):
        e = C.c_void_p()
        api.event_create(h, C.byref(e))
        ev.append(e)

    def run(n_steps, timed):
        full, rem = divmod(n_steps, HORIZON)
        for k in range(full):
            marks = timed and k < n_pairs
            if marks:
                lib.eb_event_record(ev[2 * k], sp)
            if args.open_loop:
                tape_launch()
            elif args.eager:
                eager_steps(0, HORIZON)
            else:
                rc = lib.eb_plan_launch(plan, sp)
                if rc != 0:
                    api.check(rc)
            if marks:
                lib.eb_event_record(ev[2 * k + 1], sp)
            end_of_horizon()
        eager_steps(0, rem)
        drain_gathers()
        return full

    run(args.warmup, False)
    torch.cuda.synchronize()
    # launch form of the timed region: the same 25 launches per rollout either as one hipGraph replay or as 25 host
    # calls; which one keeps the queue fuller depends on the host, so two warm rollouts of each are timed first
    if not (args.eager or args.graph or args.open_loop) and args.warmup >= HORIZON:
        trial = {False: [], True: []}
        for form in (False, True, False, True, False, True):
            args.eager = form
            run(HORIZON, False)
            lib.eb_event_record(ev[0], sp)
            run(4 * HORIZON, False)
            lib.eb_event_record(ev[1], sp)
            ms_t = C.c_float()
            api.event_elapsed_ms(ev[0], ev[1], C.byref(ms_t))
            trial[form].append(ms_t.value)
        args.eager = min(trial[True]) < min(trial[False])
        torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    full = run(args.steps, True)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    ms = C.c_float()
    ev_ms, n_marked = 0.0, min(full, n_pairs)
    for k in range(n_marked):
        api.event_elapsed_ms(ev[2 * k], ev[2 * k + 1], C.byref(ms))
        ev_ms += ms.value
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())

    if rank == 0:
        value = n_env * world * args.steps / dt_max
        alg = alg_bytes_per_env_step(n_veh) * n_env
        if n_marked:
            launch_s = ev_ms * 1e-3 / (n_marked * HORIZON)   # HIP events on the launch stream, per rollout launch
        else:
            launch_s = dt_max / args.steps                   # fewer than 25 steps: whole-region wall time
        achieved = alg / launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'r1_pmc_traffic.json')   # from a separate --pmc run (scripts/pmc_traffic.sh)
        if os.path.isfile(tpath) and n_env == N_ENV and n_veh == N_VEH:
            traffic = json.load(open(tpath)).get('hbm_bytes_per_launch')
        cfg = 'configs[2]' if (n_env, n_veh) == (N_ENV, N_VEH) else 'custom'
        kernel = 'eb::rollout_fused_4x8<0, true, float>'
        form = 'closed-loop rollout_out (one kernel launch per step, %s)' % ('eager' if args.eager else '25-launch hipGraph replays')
        if args.open_loop:
            # one launch per 25-step tape; HBM sees the initial and final obs once, actions and outputs every step
            alg = (28 * HORIZON + 72 + 32 * n_veh) * n_env
            launch_s *= HORIZON
            achieved = alg / launch_s / 1e9
            traffic = None
            kernel = 'eb::rollout_tape_4x8<0, true, float>'
            form = ('OPEN-LOOP eb_rollout_tape (the whole %d-step action tape in one launch, records and ego state in registers; '
                    'VALU-bound — reported apart from the closed-loop headline)' % HORIZON)
        line = {
            'metric': 'env-steps/s (batched rollout) at N_env x N_veh; achieved HBM GB/s vs peak',
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt_max * 1e3 / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: N_env=%d per GPU, N_veh=%d, horizon=%d, task=%s, mode=training, %s'
                                   % (cfg, n_env, n_veh, HORIZON, TASK, form),
                       'n_env_per_gpu': n_env, 'n_veh': n_veh, 'horizon': HORIZON,
                       'parallelism': 'env-shard x%d, all-gather of the 8-float episodic summary per horizon' % world},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': kernel, 'alg_bytes_per_launch': alg,
                         'avg_launch_us': launch_s * 1e6,
                         'launches_timed': n_marked * (1 if args.open_loop else HORIZON)},
            'summary': [float(x) for x in combine_summaries(gathered[0]).tolist()],
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg runs at N = 1 only
            line['cpu_baseline'] = cpu_baseline(inp, obs0.cpu().numpy(), n_veh)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    for e in ev:
        api.event_destroy(e)
    api.plan_destroy(plan)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
