#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched model rollout (EnvironmentModel.rollout_out) on MI355X.

A "step" is one rollout_out pass over the whole batch = B env-steps: ONE kernel launch per step (the closed-loop
form — a policy may sit between two steps — not the fused open-loop kernel).  Headline workload at N GPUs:
BASELINE.json configs[2] per GPU (N_env = 65 536, N_veh = 32, horizon 25, task `left`, training mode, fp32), weak
scaling: every rank owns an independent shard of envs and there is no data-path collective; the only exchange is one
all-gather (RCCL) of the 8-float episodic-return summary at the end of every horizon, inside the timed region (N = 1, one
process: no process group exists and the "gather" of the one rank's summary is a view — no launch; the summary kernels still run).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started WITHOUT a launcher and with --gpus N > 1, bench.py starts its own N ranks: it re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>`
(self_launch_argv); rank 0 still prints the one line.

The K timed steps run as rollouts of min(K, 25) launches (+ a shorter last one when 25 does not divide K), each followed
by the episodic summary (eb_episode_summary: two small kernels; --acc-summary: accumulating launches + one fold) and its all-gather; the K-step region is bracketed by barrier +
synchronize, measured `--repeats` times (default 11) and the MEDIAN is `value` (min / max are reported too).  The
launches go out either as one hipGraph replay per rollout (eb_plan_*, --graph) or as host calls (--eager); by default
a short untimed trial picks the faster form and `config.workload` names the one that ran.  Rank 0 prints ONE JSON line:

  roofline      algorithmic bytes per launch (104 + 32 N_veh per env-step) / the rollout kernel's
                average launch duration from HIP event pairs (eb_event_*) on the launch stream around every rollout of
                the timed region (inter-kernel gaps included), against the 8 TB/s HBM peak.  The headline working set
                (two 36 MB obs buffers + outputs) fits the 256 MiB Infinity Cache, so
  roofline.hbm_resident   repeats the measurement where it cannot: (a) the same kernel and batch size cycling over 8
                independent env sets (lane l steps, then lane l+1, ...: 518 MB of other traffic between a line's write
                and its re-read, 0.9 GB footprint), (b) one batch of 524 288 envs (290 MB per obs buffer), (c) the env sets
                of (a) dealt to four HIP streams, so that launches of different sets overlap (wall-clock timed);
  strong        BASELINE configs[3]: 262 144 envs in total, split 262 144 / N per rank (strong scaling; at N = 1 the
                one-GPU reference point of that curve);
  extra         configs[1] (4 096 x 16, fp32; + `one_launch_forms`: the same 25 steps as ONE gated launch with open gates /
                fed by a second stream, and the open-loop tape), configs[4] (65 536 x 64, fp16 state) and `env_step` — the
                env-side step of endtoend.py (CrossroadEnd2end.step through eb_env_step, one launch) at 65 536 and 4 096 envs
                x 16 candidates with its own algorithmic bytes and roofline fraction — on rank 0 at N = 1;
  cpu_baseline  the CPU oracle (oracle/, plain-C port of the reference path, OpenMP over envs) timed on this box's host
                cores on a bounded sample of the same workload, on rank 0 after every timed GPU region (at any N: "next to
                the reference CPU path ... in the same run"); never part of the GPU path.
  At N > 1 also: roofline.aggregate (sum over ranks of the per-rank achieved GB/s against N x 8 TB/s, with the per-rank
                launch durations' min / max) and rccl_ranks / backend = what torch.distributed reports for the job.
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
STRONG_TOTAL = 262144                            # BASELINE.json configs[3]
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s is what a float4 copy reaches)
MALL_BYTES = 256 << 20                           # Infinity Cache
MAX_PAIRS = 64                                   # HIP event pairs per repeat
TWO_PASS_SUMMARY = True                          # False: --acc-summary (A/B aid)


def alg_bytes_per_env_step(n_veh, f16=False):
    return (68 + 16 * n_veh) if f16 else (104 + 32 * n_veh)   # 1128 B at N_veh = 32 fp32


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def median(xs):
    s = sorted(xs)
    n = len(s)
    return s[n // 2] if n & 1 else 0.5 * (s[n // 2 - 1] + s[n // 2])


def self_launch_argv(argv, n_gpus, port=None):
    """The command bench.py re-executes itself with when it is started without a launcher and --gpus N > 1: one rank
    per GPU under torch.distributed.run on a free local port (the bench contract's own form)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(int(n_gpus)),
            '--master-addr', '127.0.0.1', '--master-port', str(int(port)), os.path.abspath(__file__)] + list(argv)


def gather_floats(torch, dist, values, device):
    """[world, len(values)] float64: every rank's values (all-gather on the job's backend)"""
    t = torch.tensor(values, dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.tolist()] for o in out]


def cpu_baseline(inp, obs0, n_veh, budget_s=18.0):
    """Oracle timed on host cores on a bounded sample (8192 envs x HORIZON steps, repeated until the budget is
    used): 1 thread (the reference pins TF to 1 thread) and OpenMP over envs on 8 / 32 / 64 threads where the box
    has them (a cgroup CPU quota can make more threads slower, so every count is reported and the best is `value`)."""
    from tests._helpers import HostModel, oracle_lib
    api = oracle_lib()
    api.lib.eb_oracle_set_threads.restype = C.c_int
    api.lib.eb_oracle_set_threads.argtypes = [C.c_int]
    b_cpu = 8192
    host = HostModel(api, TASK, n_veh=n_veh)
    obs, act, ref = obs0[:b_cpu].copy(), inp['actions'][:, :b_cpu].copy(), inp['ref_idx'][:b_cpu].copy()
    counts = sorted(set([1] + [c for c in (8, 32, 64) if c <= host_cores()]))
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
            'host_logical_cpus': host_cores(),
            'sample': '%d envs x %s steps on %s threads (N_veh=%d, same seeded inputs), oracle/envbuild_oracle.c, OpenMP '
                      'over envs; the 1-thread figure is the reference-faithful setting (the reference pins TF to 1 thread)'
                      % (b_cpu, '/'.join(str(v[1]) for v in res.values()), '/'.join(str(k) for k in res), n_veh)}


class Shard(object):
    """One rank's resident buffers for a (n_env, n_veh, storage) workload and the launch forms over them.

    `lanes` > 1: that many independent env sets of n_env rows each (same inputs, distinct buffers), stepped round-robin —
    lane 0 step t, lane 1 step t, ... — so that every launch streams from and to memory no other recent launch touched."""

    def __init__(self, torch, model, n_env, n_veh, seed, f16=False, lanes=1, keep_host=False, streams=1):
        from env_build_amd.synthetic import make_rollout_inputs
        self.torch, self.model, self.n_env, self.n_veh, self.f16, self.lanes = torch, model, n_env, n_veh, f16, lanes
        dev = model.device
        inp = make_rollout_inputs(model.task, n_env, n_veh, HORIZON, seed=seed)      # (the handle's task: `left` for every BASELINE config)
        ego = torch.from_numpy(inp['ego']).to(dev)
        self.ref_idx = torch.from_numpy(inp['ref_idx']).to(dev)
        trk = model.ref_path.tracking_error_vector_batched(ego[:, 3].contiguous(), ego[:, 4].contiguous(),
                                                           ego[:, 5].contiguous(), ego[:, 0].contiguous(), 0,
                                                           ref_indexes=self.ref_idx).t
        obs0 = torch.cat([ego, trk, torch.from_numpy(inp['veh']).to(dev)], 1).contiguous()
        self.obs0_f32 = obs0 if keep_host else None
        self.inp = inp if keep_host else None
        if f16:
            obs0 = obs0.to(torch.float16).contiguous()
        self.tape = torch.from_numpy(inp['actions']).to(dev)                      # [H, B, 2]
        self.obs0 = [obs0] + [obs0.clone() for _ in range(lanes - 1)]
        self.work = [torch.empty_like(obs0) for _ in range(lanes)]
        self.final = [torch.empty_like(obs0) for _ in range(lanes)]
        self.out5 = [torch.empty((HORIZON, 5, n_env), dtype=torch.float32, device=dev) for _ in range(lanes)]
        self.api, self.h, self.lib = model.api, model.handle, model.api.lib
        self.sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        # streams > 1: the lanes (independent env sets) are dealt to that many HIP streams — lane l's steps stay ordered on its
        # own stream, launches of different lanes may overlap (one set's write-back under the next set's reads)
        self._streams = [torch.cuda.Stream(device=dev) for _ in range(max(0, min(streams, lanes) - 1))]
        self.sps = [self.sp] + [C.c_void_p(st.cuda_stream) for st in self._streams]
        if self._streams:
            torch.cuda.synchronize()          # the buffers above were filled on the current stream
        self.step_fn = self.lib.eb_rollout_step_f16 if f16 else self.lib.eb_rollout_step
        self._eager, self._plans = {}, {}
        self.acc = None

    def enable_acc(self):
        """the rollout's launches become the ACCUMULATING ones (eb_rollout_step_acc, ABI 5): the episodic summary's sums are
        collected by the launches themselves and the per-horizon summary is one small fold (eb_episode_acc_finish)"""
        assert not self.f16 and self.lanes == 1
        nb = C.c_int64()
        self.api.episode_acc_bytes(self.h, self.n_env, HORIZON, C.byref(nb))
        self.acc = self.torch.empty((max(16, nb.value),), dtype=self.torch.uint8, device=self.model.device)
        self.step_fn = self.lib.eb_rollout_step_acc
        self.close()
        self._eager = {}

    def footprint_bytes(self):
        per = self.obs0[0].numel() * self.obs0[0].element_size()
        return self.lanes * (3 * per + self.out5[0].numel() * 4) + self.tape.numel() * 4

    def eager_args(self, h):
        """argument tuples of an h-step rollout, ping-ponging as eb_rollout_tape does (the last step lands in `final`)"""
        if h not in self._eager:
            p = lambda t: C.c_void_p(t.data_ptr())
            per_lane = []
            for l in range(self.lanes):
                dst = [self.final[l] if (h - 1 - t) % 2 == 0 else self.work[l] for t in range(h)]
                src = [self.obs0[l]] + dst[:-1]
                per_lane.append([(self.h, self.n_env, p(src[t]), p(self.tape[t]), p(self.ref_idx), 0, p(dst[t]),
                                  p(self.out5[l][t]), None) +
                                 ((p(self.acc), t, h, p(self.out5[l][t - 1]) if t else None) if self.acc is not None else ()) +
                                 (self.sps[l % len(self.sps)],) for t in range(h)])
            self._eager[h] = [per_lane[l][t] for t in range(h) for l in range(self.lanes)]   # round-robin over the lanes
        return self._eager[h]

    def plan(self, h):
        if self.f16 or self.lanes != 1:
            return None
        if h not in self._plans:
            p = lambda t: C.c_void_p(t.data_ptr())
            plan = C.c_void_p()
            self.api.plan_create(self.h, self.n_env, h, p(self.obs0[0]), p(self.tape), p(self.ref_idx), 0, p(self.work[0]),
                                 p(self.final[0]), p(self.out5[0]), None, None if self.acc is None else p(self.acc), C.byref(plan))
            self._plans[h] = plan
        return self._plans[h]

    def launch_rollout(self, h, eager):
        """h closed-loop steps (per lane): one kernel launch each"""
        plan = None if eager else self.plan(h)
        if plan is not None:
            rc = self.lib.eb_plan_launch(plan, self.sp)
            if rc != 0:
                self.api.check(rc)
            return
        fn = self.step_fn
        for a in self.eager_args(h):
            rc = fn(*a)
            if rc != 0:
                self.api.check(rc)

    def launch_tape(self, h):
        p = lambda t: C.c_void_p(t.data_ptr())
        fn = self.lib.eb_rollout_tape_f16 if self.f16 else self.lib.eb_rollout_tape
        rc = fn(self.h, self.n_env, h, p(self.obs0[0]), p(self.tape), p(self.ref_idx), 0, p(self.work[0]), p(self.final[0]),
                p(self.out5[0]), self.sp)
        if rc != 0:
            self.api.check(rc)

    def close(self):
        for plan in self._plans.values():
            self.api.plan_destroy(plan)
        self._plans = {}


class Timer(object):
    """K steps of a Shard as rollouts of <= HORIZON launches, event pairs around every rollout, `repeats` timed regions
    each bracketed by barrier + synchronize, max over ranks per region."""

    def __init__(self, torch, dist, use_dist, shard, with_summary):
        self.torch, self.dist, self.use_dist, self.s, self.with_summary = torch, dist, use_dist, shard, with_summary
        self.api, self.lib = shard.api, shard.lib
        self.ev = []
        for _ in range(2 * MAX_PAIRS):
            e = C.c_void_p()
            self.api.event_create(shard.h, C.byref(e))
            self.ev.append(e)
        dev = shard.model.device
        world = dist.get_world_size() if use_dist else 1
        # two summary buffers: the all-gather of rollout k overlaps the kernels of rollout k+1 (RCCL's own stream)
        self.summaries = [torch.zeros((8,), dtype=torch.float32, device=dev) for _ in range(2)]
        self.in_flight = [None, None]
        self.gathered = torch.zeros((world, 8), dtype=torch.float32, device=dev)
        self.n_rollouts = 0
        self.eager = True
        self.open_loop = False
        if with_summary and not shard.f16 and shard.lanes == 1 and not TWO_PASS_SUMMARY:
            shard.enable_acc()

    @staticmethod
    def segments(n_steps):
        full, rem = divmod(n_steps, HORIZON)
        return [HORIZON] * full + ([rem] if rem else [])

    def end_of_rollout(self, h):
        # episodic-return summary of this shard, then the only inter-GPU exchange
        from env_build_amd.sharding import gather_summaries_async
        s = self.s
        slot = self.n_rollouts & 1
        self.n_rollouts += 1
        if self.in_flight[slot] is not None:
            self.gathered = self.in_flight[slot].result()      # the gather of two rollouts ago: long finished
        p = lambda t: C.c_void_p(t.data_ptr())
        if s.acc is not None and not self.open_loop:     # the launches have collected the sums: one fold over the per-block records
            self.api.episode_acc_finish(s.h, s.n_env, h, p(s.acc), p(self.summaries[slot]), s.sp)
        else:
            self.api.episode_summary(s.h, s.n_env, h, p(s.out5[0]), p(s.final[0]), p(self.summaries[slot]), s.sp)
        # env_build_amd/sharding.py: 8 floats per rank.  (copy=False: this timer owns two summary buffers and reads a result before its
        # buffer is written again — at world == 1 the "gather" is then a view, no launch; at N > 1 it is RCCL's all-gather)
        self.in_flight[slot] = gather_summaries_async(self.summaries[slot], copy=False)

    def drain(self):
        for slot in (self.n_rollouts & 1, (self.n_rollouts + 1) & 1):      # oldest first
            if self.in_flight[slot] is not None:
                self.gathered = self.in_flight[slot].result()
                self.in_flight[slot] = None

    def run(self, n_steps, marks):
        segs = self.segments(n_steps)
        for k, h in enumerate(segs):
            m = marks and k < MAX_PAIRS
            if m:
                self.lib.eb_event_record(self.ev[2 * k], self.s.sp)
            if self.open_loop:
                self.s.launch_tape(h)
            else:
                self.s.launch_rollout(h, self.eager)
            if m:
                self.lib.eb_event_record(self.ev[2 * k + 1], self.s.sp)
            if self.with_summary:
                self.end_of_rollout(h)
        if self.with_summary:
            self.drain()
        return segs

    def pick_form(self):
        """graph replay or host calls: whichever keeps the queue fuller on this host (untimed trial, three rounds each)"""
        if self.s.plan(HORIZON) is None:
            self.eager = True
            return
        trial = {False: [], True: []}
        ms = C.c_float()
        for form in (False, True, False, True, False, True):
            self.eager = form
            self.run(HORIZON, False)
            self.lib.eb_event_record(self.ev[0], self.s.sp)
            self.run(4 * HORIZON, False)
            self.lib.eb_event_record(self.ev[1], self.s.sp)
            self.api.event_elapsed_ms(self.ev[0], self.ev[1], C.byref(ms))
            trial[form].append(ms.value)
        self.eager = min(trial[True]) < min(trial[False])
        self.torch.cuda.synchronize()

    def measure(self, n_steps, warmup, repeats):
        """-> dict(dt = [per-repeat seconds, max over ranks], launch_us = event-timed duration per launch: the mean within a repeat,
        the median over the repeats (the statistic of `value`), launch_us_by_repeat, launches_timed)"""
        torch, dist = self.torch, self.dist
        self.run(warmup, False)
        # the launch forms of the K-step region are built before it is timed (argument tuples / hipGraph of a rollout of this
        # length: host work, no steps) — a region that is shorter than the warm-up's rollouts would otherwise build them on the clock
        for h in set(self.segments(n_steps)):
            if not self.open_loop:
                self.s.eager_args(h)
                if not self.eager:
                    self.s.plan(h)
        torch.cuda.synchronize()
        dts, per_repeat, ev_launches = [], [], 0
        ms = C.c_float()
        for _ in range(repeats):
            if self.use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            segs = self.run(n_steps, True)
            torch.cuda.synchronize()
            if self.use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
            ev_ms, n_l = 0.0, 0
            for k, h in enumerate(segs[:MAX_PAIRS]):
                self.api.event_elapsed_ms(self.ev[2 * k], self.ev[2 * k + 1], C.byref(ms))
                ev_ms += ms.value
                n_l += (1 if self.open_loop else h * self.s.lanes)
            per_repeat.append(ev_ms * 1e3 / max(1, n_l))
            ev_launches += n_l
        t = torch.tensor(dts, dtype=torch.float64, device=self.s.model.device if not self.use_dist or dist.get_backend() == 'nccl' else 'cpu')
        if self.use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return dict(dt=[float(x) for x in t.tolist()], launch_us=sorted(per_repeat)[len(per_repeat) // 2], launch_us_by_repeat=per_repeat,
                    launches_timed=ev_launches)

    def close(self):
        for e in self.ev:
            self.api.event_destroy(e)
        self.s.close()


def pmc_traffic(which, *path):
    """HBM bytes per launch from the newest profiles/r*_pmc_traffic.json WHOSE KERNEL SOURCES ARE THE ONES THIS RUN EXECUTES (the
    file records build.kernel_hash() of the code it measured): -> (bytes or None, source note).  A file measured on other code is
    refused, not silently reused — the counters of a changed kernel are unknown until scripts/pmc_traffic.sh has run again."""
    import glob
    from env_build_amd import build as _build
    want = _build.kernel_hash(which)
    stale = []
    for tpath in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), reverse=True):
        d = json.load(open(tpath))
        name = os.path.basename(tpath)
        if (d.get('kernel_source_hash') or {}).get(which) != want:
            stale.append(name)
            continue
        v = d
        for k in path:
            v = (v or {}).get(k)
        if v is not None:
            return v, ('profiles/%s: FETCH_SIZE x 2 + WRITE_SIZE from separate rocprofv3 --pmc passes of this kernel (same sources: '
                       'hash %s...), not measured in this run' % (name, want[:12]))
    return None, ('no PMC traffic file for the kernel sources of this run (hash %s...; measured on other code: %s) — run '
                  'scripts/pmc_traffic.sh' % (want[:12], ', '.join(stale) or 'none'))


def roofline_of(alg_bytes, launch_us):
    achieved = alg_bytes / (launch_us * 1e-6) / 1e9
    return achieved, achieved / HBM_PEAK_GBS


def side_config(torch, dist, model, n_env, n_veh, seed, steps, warmup, repeats, f16=False, lanes=1, use_dist=False,
                forms=('graph', 'eager'), streams=1, with_summary=False, tile=None):
    """One more workload through the same protocol (with_summary: incl. the episodic summary kernels and their gather per
    horizon, as the headline; tile: eb_debug_set_tile variant for this measurement only): -> compact result dict."""
    if tile is not None:
        model.api.debug_set_tile(model.handle, tile)
    shard = Shard(torch, model, n_env, n_veh, seed, f16=f16, lanes=lanes, streams=streams)
    tm = Timer(torch, dist, use_dist, shard, with_summary=with_summary)
    tm.run(min(warmup, HORIZON), False)
    torch.cuda.synchronize()
    if 'graph' in forms and 'eager' in forms:
        tm.pick_form()
    else:
        tm.eager = 'graph' not in forms or shard.plan(HORIZON) is None
    r = tm.measure(steps, warmup, repeats)
    world = dist.get_world_size() if use_dist else 1
    dt = median(r['dt'])
    alg = alg_bytes_per_env_step(n_veh, f16) * n_env
    if len(shard.sps) > 1:      # event pairs sit on one stream only: the per-launch time of a multi-stream run is wall clock / launches
        r['launch_us'] = dt * 1e6 / (steps * lanes)
    achieved, frac = roofline_of(alg, r['launch_us'])
    out = {'n_env_per_gpu': n_env, 'n_veh': n_veh, 'dtype': 'f16 state / f32 arithmetic' if f16 else 'f32', 'lanes': lanes,
           'value': n_env * lanes * world * steps / dt, 'unit': 'env-steps/s',
           'ms_per_step': dt * 1e3 / (steps * lanes), 'steps': steps, 'repeats': len(r['dt']),
           'launch_form': 'eager' if (tm.eager or shard.plan(HORIZON) is None) else 'hipGraph',
           'alg_bytes_per_launch': alg, 'avg_launch_us': r['launch_us'], 'launches_timed': r['launches_timed'],
           'achieved_GBs': achieved, 'frac': frac, 'footprint_MB': shard.footprint_bytes() / 1e6}
    if len(shard.sps) > 1:
        out.update(streams=len(shard.sps), timing='wall clock of the timed region / launches (the lanes run on %d HIP streams)' % len(shard.sps))
    if f16 and (n_env, n_veh, lanes) == (N_ENV, 64, 1):      # configs[4]: HBM bytes from the PMC passes of scripts/pmc_traffic.sh
        out['traffic'], out['traffic_source'] = pmc_traffic('rollout', 'fp16_x64', 'hbm_bytes_per_launch')
    if with_summary:
        out['protocol'] = ('episodic summary kernels + their gather once per horizon inside the timed region, as the headline '
                           '(one process: the gather of one rank\'s summary is a view, no launch)')
    if tile is not None:
        out['tile_variant'] = tile
        model.api.debug_set_tile(model.handle, -1)
    tm.close()
    del shard, tm
    torch.cuda.empty_cache()
    return out


def facade_rollout_bench(torch, EnvironmentModel, dev, n_env, n_veh, seed, n_calls=1000):
    """The drop-in call itself: `EnvironmentModel.rollout_out(actions)` (DAM:118-126; the reference's callers loop over it,
    hier_decision.py:91-96) — microseconds per call over `n_calls` back-to-back calls in 25-step episodes, host time (the loop's own
    duration: what the caller's thread spends) and with the drain (+ the final synchronize: the rate the device sustains), for both
    output modes, next to the same loop through the raw C entry eb_rollout_step over two fixed buffers."""
    from env_build_amd.synthetic import make_rollout_inputs
    inp = make_rollout_inputs(TASK, n_env, n_veh, HORIZON, seed=seed)
    tape = torch.from_numpy(inp['actions']).to(dev)
    ref = torch.from_numpy(inp['ref_idx']).to(dev)
    out = {'workload': 'EnvironmentModel.rollout_out, N_env=%d, N_veh=%d, %d calls in %d-step episodes (reset(obses, ref_indexes) '
                       'between them), task %s, training mode' % (n_env, n_veh, n_calls, HORIZON, TASK), 'unit': 'us per call'}
    obs0 = None
    for copy in (True, False):
        m = EnvironmentModel(TASK, num_future_data=0, mode='training', n_veh=n_veh, device=dev, copy_outputs=copy)
        if obs0 is None:
            ego = torch.from_numpy(inp['ego']).to(dev)
            trk = m.ref_path.tracking_error_vector_batched(ego[:, 3].contiguous(), ego[:, 4].contiguous(), ego[:, 5].contiguous(),
                                                           ego[:, 0].contiguous(), 0, ref_indexes=ref).t
            obs0 = torch.cat([ego, trk, torch.from_numpy(inp['veh']).to(dev)], 1).contiguous()
        acts = [tape[t] for t in range(HORIZON)]          # (the views are made once: the caller's policy output stands for them)
        runs = []
        for rep in range(3):
            m.reset(obs0, ref)
            for t in range(HORIZON):
                m.rollout_out(acts[t])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_calls):
                if i % HORIZON == 0:
                    m.reset(obs0, ref)
                r = m.rollout_out(acts[i % HORIZON])
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            runs.append(((t1 - t0) / n_calls * 1e6, (t2 - t0) / n_calls * 1e6))
        del r
        runs.sort(key=lambda x: x[1])
        out['copy_outputs=%s' % copy] = {'host_us': runs[1][0], 'with_drain_us': runs[1][1], 'runs': runs}
        api, lib, h = m.api, m.api.lib, m.handle
    # the raw C entry over two fixed obs buffers: what the facade wraps
    p = lambda t: C.c_void_p(t.data_ptr())
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bufs = [torch.empty_like(obs0), torch.empty_like(obs0)]
    o5, sc = torch.empty((5, n_env), device=dev), torch.empty((n_env, 2), device=dev)
    args = []
    for t in range(HORIZON):
        src = obs0 if t == 0 else bufs[(t - 1) & 1]
        args.append((h, n_env, p(src), p(tape[t]), p(ref), 0, p(bufs[t & 1]), p(o5), p(sc), sp))
    runs = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_calls):
            rc = lib.eb_rollout_step(*args[i % HORIZON])
            if rc != 0:
                api.check(rc)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        runs.append(((t1 - t0) / n_calls * 1e6, (t2 - t0) / n_calls * 1e6))
    runs.sort(key=lambda x: x[1])
    out['c_abi_eb_rollout_step'] = {'host_us': runs[1][0], 'with_drain_us': runs[1][1], 'runs': runs}
    base = out['c_abi_eb_rollout_step']['with_drain_us']
    out['facade_over_c_abi'] = {k: out[k]['with_drain_us'] / base for k in ('copy_outputs=True', 'copy_outputs=False')}
    del m
    torch.cuda.empty_cache()
    return out


def one_launch_forms(torch, model, n_env, n_veh, seed, reps=100):
    """configs[1] is launch-bound as one kernel per step; the same H = 25 closed-loop-capable rollout in ONE launch:
    eb_rollout_gated (a device-side gate in front of every step) with every gate open and the states published after
    every step, the same fed step by step by a producer kernel on a second stream (eb_gate_feed: what a policy kernel
    in the loop does, minus the policy), and — for reference — the open-loop tape kernel.  Wall clock over `reps`
    back-to-back rollouts, bracketed by synchronize."""
    from env_build_amd.synthetic import make_rollout_inputs
    dev, H = model.device, HORIZON
    api, lib, h = model.api, model.api.lib, model.handle
    inp = make_rollout_inputs(TASK, n_env, n_veh, H, seed=seed)
    ego = torch.from_numpy(inp['ego']).to(dev)
    ref = torch.from_numpy(inp['ref_idx']).to(dev)
    trk = model.ref_path.tracking_error_vector_batched(ego[:, 3].contiguous(), ego[:, 4].contiguous(), ego[:, 5].contiguous(),
                                                       ego[:, 0].contiguous(), 0, ref_indexes=ref).t
    obs0 = torch.cat([ego, trk, torch.from_numpy(inp['veh']).to(dev)], 1).contiguous()
    tape = torch.from_numpy(inp['actions']).to(dev)
    live = torch.empty_like(tape)
    work, out = torch.empty_like(obs0), torch.empty_like(obs0)
    out5 = torch.empty((H, 5, n_env), device=dev)
    steps = torch.empty((H,) + tuple(obs0.shape), device=dev)
    nb = C.c_int32()
    api.rollout_gated_blocks(h, n_env, C.byref(nb))
    nb = nb.value
    if nb == 0:
        return None
    i32 = dict(dtype=torch.int32, device=dev)
    open_gates, status = torch.ones(H, **i32), torch.zeros(2, **i32)
    p = lambda t: C.c_void_p(t.data_ptr())
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    spin = 1 << 20
    k = [0]

    def gated_open():
        api.rollout_gated(h, n_env, H, p(obs0), p(tape), p(ref), 0, p(work), p(out), p(out5), p(steps), p(open_gates), p(done[0]),
                          nb, p(status), spin, sp)

    def fed(wait_after):
        def run():
            i = k[0] % R
            k[0] += 1
            api.gate_feed(h, n_env, H, nb, p(tape), p(live), p(ready[i]), p(done[i]), p(status), spin, sp, wait_after, None)
            api.rollout_gated(h, n_env, H, p(obs0), p(live), p(ref), 0, p(work), p(out), p(out5), p(steps), p(ready[i]), p(done[i]),
                              nb, p(status), spin, sp)
        return run

    def tape_kernel():
        api.rollout_tape(h, n_env, H, p(obs0), p(tape), p(ref), 0, p(work), p(out), p(out5), sp)

    runs = 5
    R = reps + 8             # one set of flags per fed repetition; zeroed again before every run
    ready, done = torch.zeros((R, H), **i32), torch.zeros((R, H, nb, 16), **i32)
    res = {'horizon': H, 'gated_blocks': nb, 'rollouts_timed': reps, 'runs': runs, 'statistic': 'us_per_step = median of the runs (us_per_step_min: the fastest)'}
    # fed by a second stream, two orderings of the producer: (a) wait_after = 0 — the staged tape and the zeroed flags were ready long
    # ago, so the feed of rollout k+1 may start while rollout k is still running; (b) wait_after = 1 — eb_gate_feed orders the feed
    # behind everything the caller has enqueued on its stream (an event record + a stream wait per feed), which includes the previous
    # gated rollout: the rollouts serialise.  Round 3 reported 6.7 us (before the ordering existed) and 10.1 us (after) under one name.
    for name, fn in (('gated_open_gates', gated_open), ('gated_fed_by_second_stream', fed(0)),
                     ('gated_fed_ordered_behind_the_caller_stream', fed(1)), ('open_loop_tape', tape_kernel)):
        us = []
        for _ in range(runs):
            if name.startswith('gated_fed'):
                k[0] = 0
                ready.zero_(); done.zero_()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            us.append((time.perf_counter() - t0) / reps * 1e6 / H)
        res[name] = {'us_per_step': median(us), 'us_per_step_min': min(us), 'us_per_step_max': max(us),
                     'value': n_env / (median(us) * 1e-6), 'unit': 'env-steps/s'}
    if status.cpu().tolist() != [0, 0]:
        raise RuntimeError('gated rollout gave up at a gate: status %s' % status.cpu().tolist())
    return res


def env_step_alg_bytes(D, m_cand):
    """Algorithmic bytes of one CrossroadEnd2end.step per env (env-side step): read obs 4D + ego 24 + raw action 8
    + ref index 4 + candidates 16M + their modes M; write obs 4D + ego 24 + params 16 + scaled action 8 + candidates 16M +
    five outputs 20 + done code 1  ->  8D + 32M + M + 105  (961 B at D = 41, M = 16).  The 16-term reward dict (64 B) is
    optional in the C-ABI and not requested here; path tables, slot modes and the re-entry table are L2-resident."""
    return 8 * D + 33 * m_cand + 105


def env_step_bench(torch, dev, n_env, n_cand=16, seg=10, reps=40, tile=None, waves=0, by_progress=None):
    """The env-side step (E2E:132-144 = action scaling, reward, ego step, traffic step, observation,
    done code, pool re-entry) through the raw C entry eb_env_step: pre-allocated ping-pong observation buffers, no Python
    allocation in the loop.  `reps` segments of `seg` steps from the same reset state (restored between segments, outside
    the event pairs, so that the egos stay on the map as they do under a driver that resets finished envs), mild random
    actions; HIP event pairs around every segment."""
    from env_build_amd import _capi
    from env_build_amd.endtoend import CrossroadEnd2end
    env = CrossroadEnd2end(TASK, n_env=n_env, multi_display=True, traffic='pool', n_cand=n_cand, device=dev)
    env.seed(0)
    env.reset()
    api, lib = env.api, env.api.lib
    if tile is not None:      # tuning aids (scripts/sweep_env_tile.py): eb_debug_set_tile / eb_debug_set_env_waves on the env's handle
        api.debug_set_tile(env._h, int(tile))
    if waves:
        api.debug_set_env_waves(env._h, int(waves))
    if by_progress is not None:   # A/B aid: the step kernel's issue priority by phase forced on / off (eb_debug_set_rollout_sched)
        api.debug_set_rollout_sched(env._h, -1, int(by_progress))
    B, M, D = n_env, env.n_cand, env.obs_dim
    g = torch.Generator(device='cpu').manual_seed(3)
    tape = torch.stack([torch.rand((seg, B), generator=g) * 0.6 - 0.3, torch.rand((seg, B), generator=g) * 0.8 - 0.2], 2).to(dev).contiguous()
    keep = {k: getattr(env, k).clone() for k in ('_obs', '_ego', '_params', '_cand')}
    obs = [torch.empty_like(env._obs) for _ in range(2)]
    ego, params, cand = env._ego, env._params, env._cand.contiguous()
    f32 = dict(dtype=torch.float32, device=dev)
    scaled, out5, code = torch.empty((B, 2), **f32), torch.empty((5, B), **f32), torch.empty((B,), dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rule = _capi.EbRespawn(env._entry5.data_ptr(), 65.0, env.POOL_EDGE_SPAN, 8.0, 12345, 0)    # the facade's own re-entry rule
    fn = lib.eb_env_step
    h, ht = env._h, env._traffic.h
    argsets = []
    for t in range(seg):
        src, dst = obs[t & 1], obs[(t + 1) & 1]
        argsets.append((h, ht, B, p(src), p(tape[t]), p(env._ref_idx), 0, p(ego), p(params), M, p(cand), p(env._cand_mode), None,
                        p(env._v_light), p(env._virtual), p(scaled), p(out5), None, p(dst), p(code), C.byref(rule), None, None, None, sp))
    ev = []
    for _ in range(2 * reps):
        e = C.c_void_p()
        api.event_create(h, C.byref(e))
        ev.append(e)

    def restore():
        obs[0].copy_(keep['_obs']); ego.copy_(keep['_ego']); params.copy_(keep['_params']); cand.copy_(keep['_cand'])

    def segment(k):
        for a in argsets:
            rule.counter = k = k + 1
            rc = fn(*a)
            if rc != 0:
                api.check(rc)
        return k

    import gc
    gc.collect()              # handles and buffers of earlier measurements go now, not inside a timed segment
    k = 0
    for _ in range(3):
        restore(); k = segment(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        restore()
        lib.eb_event_record(ev[2 * r], sp)
        k = segment(k)
        lib.eb_event_record(ev[2 * r + 1], sp)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms, seg_us = C.c_float(), []
    for r in range(reps):
        api.event_elapsed_ms(ev[2 * r], ev[2 * r + 1], C.byref(ms))
        seg_us.append(ms.value * 1e3 / seg)
    # the masked reset a vectorised driver issues after every step (eb_env_reset_pool, one launch: draws, pool re-entry clear of
    # the new ego, reset observation, flag swap, the other rows carried over into fresh arrays): 2 % of the envs, 100 calls
    restore()
    gmask = (torch.rand((B,), generator=g) < 0.02).to(torch.uint8).to(dev)
    rrule = _capi.EbRespawn(env._entry5.data_ptr(), 0.0, 60.0, 8.0, 777, 0, env.POOL_EDGE_SPAN)
    code2 = torch.empty_like(code)
    n_reset = 100

    def resets(k0):
        for k in range(k0, k0 + n_reset):
            rrule.counter = k
            rc = lib.eb_env_reset_pool(h, ht, B, p(gmask), C.c_uint64(99), C.c_uint64(k), 1, p(ego), p(params), p(env._ref_idx), p(env._virtual),
                                       p(env._v_light), p(code2), None, M, p(cand), p(env._cand_mode), C.byref(rrule), p(obs[1]), p(obs[0]), p(code), sp)
            if rc != 0:
                api.check(rc)
    resets(1)
    lib.eb_event_record(ev[0], sp)
    resets(1 + n_reset)
    lib.eb_event_record(ev[1], sp)
    torch.cuda.synchronize()
    api.event_elapsed_ms(ev[0], ev[1], C.byref(ms))
    reset_us = ms.value * 1e3 / n_reset
    # step AND the reset of the envs it finishes as ONE launch (ABI 4, eb_env_step(auto_reset); env_step_kernel<.., AUTO>): the
    # self-sustaining loop of a vectorised driver — no state restore between segments, finished envs restart inside the launch
    restore()
    final = torch.empty_like(obs[0])
    ar = _capi.EbAutoReset(4242, 0, 1, env._ref_idx.data_ptr(), env._virtual.data_ptr(), env._v_light.data_ptr(), rrule, final.data_ptr())
    auto_sets = [a[:-4] + (C.byref(ar), None, None, sp) for a in argsets]

    def auto_segment(k):
        for a in auto_sets:
            rule.counter = ar.counter = ar.pool.counter = k = k + 1
            rc = fn(*a)
            if rc != 0:
                api.check(rc)
        return k
    k = 10000
    fin_count = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(3):                   # untimed: the loop settles into its steady done rate, which is counted here
        for a in auto_sets:
            rule.counter = ar.counter = ar.pool.counter = k = k + 1
            fn(*a)
            fin_count += (code != 0).sum()
    auto_done_rate = float(fin_count.item()) / (3 * seg * B)
    auto_us = []
    for r in range(reps):
        lib.eb_event_record(ev[2 * r], sp)
        k = auto_segment(k)
        lib.eb_event_record(ev[2 * r + 1], sp)
    torch.cuda.synchronize()
    for r in range(reps):
        api.event_elapsed_ms(ev[2 * r], ev[2 * r + 1], C.byref(ms))
        auto_us.append(ms.value * 1e3 / seg)
    for e in ev:
        api.event_destroy(e)
    us = median(seg_us)       # per-step time of the median segment: one host hiccup (a collection, a free) does not move it
    alg = env_step_alg_bytes(D, M) * B
    achieved, frac = roofline_of(alg, us)
    done_frac = float((code != 0).float().mean().item())
    traffic, traffic_src = (pmc_traffic('env_step', 'env_step', 'hbm_bytes_per_launch') if (B, M) == (N_ENV, 16) else (None, None))
    return {'workload': 'env_step: CrossroadEnd2end.step for N_env=%d single-ego envs x %d traffic candidates (task %s, D=%d): eb_env_step '
                        '= ONE launch (action scaling, reward, ego step, traffic step, observation, done code, pool re-entry)' % (B, M, TASK, D),
            'n_env_per_gpu': B, 'n_cand': M, 'obs_dim': D, 'dtype': 'f32', 'value': B / (us * 1e-6), 'unit': 'env-steps/s',
            'avg_launch_us': us, 'launches_timed': reps * seg, 'segments': {'n': reps, 'steps_each': seg, 'statistic': 'median',
                                                                      'us_per_step_min': min(seg_us), 'us_per_step_max': max(seg_us)},
            'wall_us_per_step_incl_state_restores': wall * 1e6 / (reps * seg),
            'alg_bytes_per_env_step': env_step_alg_bytes(D, M), 'alg_bytes_per_launch': alg, 'achieved_GBs': achieved, 'frac': frac,
            'traffic': traffic, 'traffic_source': traffic_src,
            'kernel': 'eb::env_step_kernel<0, %d, false, false, %d>' % ((16, 8) if B <= 1024 else (32, 8) if B <= 20480 else (64, 4)),
            'done_fraction_after_segment': done_frac,
            'masked_reset': {'entry': 'eb_env_reset_pool (one launch: eb::env_reset_pool_kernel)', 'mask_fraction': 0.02, 'calls_timed': n_reset,
                             'us_per_call': reset_us, 'carries_over': 'observation and done-code rows of the other envs (obs_src / done_src)'},
            'step_with_auto_reset': {'entry': 'eb_env_step(auto_reset) — ONE launch (eb::env_step_kernel<.., AUTO>): the step, the terminal rows '
                                              'to final_obs and the reset of the envs it finished (draws, pool re-entry, reset observation, flag swap)',
                                     'us_per_step': median(auto_us), 'us_per_step_min': min(auto_us), 'us_per_step_max': max(auto_us),
                                     'launches_timed': reps * seg, 'finished_per_step_fraction': auto_done_rate,
                                     'two_launch_equivalent_us': us + reset_us,
                                     'value': B / (median(auto_us) * 1e-6), 'unit': 'env-steps/s',
                                     'frac': roofline_of(alg, median(auto_us))[1]}}


def env_step_flows_bench(torch, dev, n_env, per_route=5, seg=10, reps=24, by_progress=None):
    """The env-side step over the SUMO-free FLOW traffic source (12 routes x per_route slots = 60 candidates per env; traffic.py,
    sumo_files/cross.rou.xml:18-44) through the raw C entry: (a) eb_env_step(flow) — the step with the flow rule in its launch — in
    short segments from a restored state, (b) eb_env_step(flow + auto_reset), ABI 5 — the self-sustaining loop: the step, the flow
    rule AND the reset of the envs it finished (eb_env_reset + the flow source's init_traffic + reset observation) as ONE launch.
    Algorithmic bytes per env-step: the step's 8 D + 33 M + 105 (env_step_alg_bytes: 2 413 B at D = 41, M = 60) + the flow rule's
    bookkeeping — active flags read + written 2 M, mode bytes rewritten M, twelve timers read + written 96, clock 8 — = 3 M + 104."""
    from env_build_amd.endtoend import CrossroadEnd2end
    from env_build_amd.traffic import RESET_SALT
    env = CrossroadEnd2end(TASK, n_env=n_env, multi_display=True, traffic='flows', per_route=per_route, auto_reset=True,
                           copy_outputs=False, device=dev)
    env.seed(0)
    env.reset()
    api, lib, fl = env.api, env.api.lib, env._flows
    if by_progress is not None:   # A/B aid, as env_step_bench
        api.debug_set_rollout_sched(env._h, -1, int(by_progress))
    B, M, D = n_env, env.n_cand, env.obs_dim
    g = torch.Generator(device='cpu').manual_seed(3)
    tape = torch.stack([torch.rand((seg, B), generator=g) * 0.6 - 0.3, torch.rand((seg, B), generator=g) * 0.8 - 0.2], 2).to(dev).contiguous()
    for t in range(60):                   # a junction in mid-traffic (the source starts with what the approach lanes hold)
        env.step(tape[t % seg])
    torch.cuda.synchronize()
    state = dict(_obs=env._obs, _ego=env._ego, _params=env._params, _cand=env._cand, active=fl.active, timer=fl.timer, emitted=fl.emitted,
                 sim_step=fl.sim_step, _mode=fl._mode, _vlight=fl._vlight, _ref=env._ref_idx, _virt=env._virtual)
    keep = {k: v.clone() for k, v in state.items()}
    obs = [env._obs.clone(), torch.empty_like(env._obs)]
    f32 = dict(dtype=torch.float32, device=dev)
    scaled, out5, code = torch.empty((B, 2), **f32), torch.empty((5, B), **f32), torch.empty((B,), dtype=torch.uint8, device=dev)
    final = torch.empty_like(obs[0])
    p = lambda t: C.c_void_p(t.data_ptr())
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rule, ar = fl.step_rule(), env._auto_rule
    ar.final_obs, ar.v_light = final.data_ptr(), fl._vlight.data_ptr()
    lw = fl.cand_lw()
    h, ht, fn = env._h, env._traffic.h, lib.eb_env_step
    base = lambda t, auto: (h, ht, B, p(obs[t & 1]), p(tape[t]), p(env._ref_idx), 0, p(env._ego), p(env._params), M, p(env._cand), p(fl._mode),
                            p(lw), p(fl._vlight), p(env._virtual), p(scaled), p(out5), None, p(obs[(t + 1) & 1]), p(code), None,
                            C.byref(ar) if auto else None, C.byref(rule), None, sp)
    sets = {a: [base(t, a) for t in range(seg)] for a in (False, True)}
    ev = []
    for _ in range(2 * reps):
        e = C.c_void_p()
        api.event_create(h, C.byref(e))
        ev.append(e)

    def restore():
        for k, v in state.items():
            v.copy_(keep[k])
        obs[0].copy_(keep['_obs'])

    def segment(k, auto):
        for a in sets[auto]:
            k += 1
            rule.counter = k
            ar.seed, ar.counter, ar.flow_seed, ar.flow_counter = 4242, k, fl.seed ^ RESET_SALT, k
            rc = fn(*a)
            if rc != 0:
                api.check(rc)
        return k

    ms, res, k = C.c_float(), {}, 100000
    for auto in (False, True):
        for _ in range(3):
            restore(); k = segment(k, auto)
        torch.cuda.synchronize()
        for r in range(reps):
            if not auto:
                restore()               # (the self-sustaining loop needs none: finished envs restart inside the launch)
            lib.eb_event_record(ev[2 * r], sp)
            k = segment(k, auto)
            lib.eb_event_record(ev[2 * r + 1], sp)
        torch.cuda.synchronize()
        us = []
        for r in range(reps):
            api.event_elapsed_ms(ev[2 * r], ev[2 * r + 1], C.byref(ms))
            us.append(ms.value * 1e3 / seg)
        res[auto] = us
    for e in ev:
        api.event_destroy(e)
    fin = float((code != 0).float().mean().item())
    step_b, rule_b = env_step_alg_bytes(D, M), 3 * M + 104
    alg = (step_b + rule_b) * B
    us0, us1 = median(res[False]), median(res[True])
    out = {'workload': 'env_step_flows: CrossroadEnd2end.step over the flow traffic source, N_env=%d x %d candidates (12 routes x %d slots; task %s, '
                       'D=%d): eb_env_step(flow) = ONE launch (the step + the flow rule: exits, accelerations, emissions, mode bytes, clock, light)'
                       % (B, M, per_route, TASK, D),
           'n_env_per_gpu': B, 'n_cand': M, 'obs_dim': D, 'dtype': 'f32', 'value': B / (us0 * 1e-6), 'unit': 'env-steps/s',
           'avg_launch_us': us0, 'launches_timed': reps * seg,
           'segments': {'n': reps, 'steps_each': seg, 'statistic': 'median', 'us_per_step_min': min(res[False]), 'us_per_step_max': max(res[False])},
           'alg_bytes_per_env_step': step_b + rule_b, 'alg_bytes_step': step_b, 'alg_bytes_flow_rule': rule_b, 'alg_bytes_per_launch': alg,
           'achieved_GBs': roofline_of(alg, us0)[0], 'frac': roofline_of(alg, us0)[1],
           'step_with_auto_reset': {'entry': 'eb_env_step(flow + auto_reset), ABI 5 — ONE launch: the step, the flow rule, the terminal rows to '
                                             'final_obs and the reset of the envs it finished (eb_env_reset, the flow source\'s init_traffic, reset '
                                             'observation, flag swap)',
                                    'us_per_step': us1, 'us_per_step_min': min(res[True]), 'us_per_step_max': max(res[True]),
                                    'launches_timed': reps * seg, 'finished_fraction_last_step': fin,
                                    'value': B / (us1 * 1e-6), 'unit': 'env-steps/s', 'frac': roofline_of(alg, us1)[1]}}
    del env
    torch.cuda.empty_cache()
    return out


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
    ap.add_argument('--repeats', type=int, default=11, help='timed regions of --steps steps each; the median is reported')
    ap.add_argument('--n-env', type=int, default=N_ENV, help='envs per GPU (default: configs[2])')
    ap.add_argument('--n-veh', type=int, default=N_VEH)
    ap.add_argument('--eager', action='store_true', help='one host launch per step instead of hipGraph replays')
    ap.add_argument('--graph', action='store_true', help='hipGraph replays (default: whichever of the two an untimed trial finds faster)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget-s', type=float, default=18.0, help='seconds of host time the cpu_baseline leg may use')
    ap.add_argument('--no-side', action='store_true', help='skip hbm_resident / strong / extra (headline line only)')
    ap.add_argument('--acc-summary', action='store_true',
                    help='A/B aid: the accumulating launches (eb_rollout_step_acc, ABI 5) + one fold per horizon instead of plain launches '
                         '+ eb_episode_summary — same-box A/B, round 5 (profiles/r5_ab_acc_summary.txt): the in-launch accumulation costs '
                         '0.2 us per launch, more than the second pass over out5 it removes (0.1 us per step slower overall)')
    ap.add_argument('--open-loop', action='store_true',
                    help='SEPARATE figure (SURVEY.md §8(f)1): eb_rollout_tape, the whole 25-step tape in one launch with the '
                         'state in registers — VALU-bound, not the HBM-bound closed-loop headline')
    ap.add_argument('--env-step', action='store_true', help='only the env-side step entries of `extra` (profiling aid), as JSON lines')
    ap.add_argument('--facade', action='store_true', help='only `extra.facade_rollout_out` (the drop-in call, host cost included), as JSON lines')
    ap.add_argument('--shield', action='store_true',
                    help='SEPARATE figure (SURVEY.md §8(f)2): eb_shield_is_safe — 5 x [policy MLP (137 -> 256 -> 256 -> 4, ELU) -> '
                         'rollout step] per start state; a "step" is one shield pass over the batch; N = 1 only')
    args = ap.parse_args()
    global TWO_PASS_SUMMARY
    TWO_PASS_SUMMARY = not args.acc_summary
    n_env, n_veh = args.n_env, args.n_veh
    if args.steps < 1 or args.repeats < 1:
        raise SystemExit('--steps and --repeats must be >= 1')
    if args.shield:
        return shield_bench(args)
    if args.env_step:
        import torch
        for b in (N_ENV, 4096):
            print(json.dumps(env_step_bench(torch, torch.device('cuda', 0), b)))
        print(json.dumps(env_step_flows_bench(torch, torch.device('cuda', 0), N_ENV)))
        return
    if args.facade:
        import torch
        from env_build_amd.dynamics_and_models import EnvironmentModel
        for b, nv, sd in ((N_ENV, N_VEH, 21), (4096, 16, 22)):
            print(json.dumps(facade_rollout_bench(torch, EnvironmentModel, torch.device('cuda', 0), b, nv, sd)))
        return

    import torch
    import torch.distributed as dist
    from env_build_amd.dynamics_and_models import EnvironmentModel
    from env_build_amd.sharding import combine_summaries

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        # started without a launcher: become the launcher (one rank per GPU over RCCL; rank 0 prints the line)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        cmd = self_launch_argv(sys.argv[1:], args.gpus)
        sys.stdout.flush()
        sys.stderr.flush()
        os.execv(cmd[0], cmd)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (the launcher started a different number of ranks)' % (args.gpus, world))
    # Test aids for a 1-GPU box: EB_BENCH_DEVICE pins every rank to one device index and EB_BENCH_BACKEND=gloo replaces
    # RCCL (which refuses two ranks on one GPU) — the N > 1 control flow (sharding, barriers, max over ranks, strong split)
    # then runs end to end with the ranks time-sharing the GPU; the numbers of such a run mean nothing.
    dev_index = int(os.environ.get('EB_BENCH_DEVICE', local_rank))
    backend = os.environ.get('EB_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    use_dist = world > 1 or os.environ.get('EB_BENCH_FORCE_DIST') == '1'   # the latter: exercise the RCCL calls on one GPU
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    # ---- headline: this rank's synthetic shard (seed = rank: independent envs per GPU), resident in HBM ----
    model = EnvironmentModel(TASK, num_future_data=0, mode='training', n_veh=n_veh, device=dev)
    shard = Shard(torch, model, n_env, n_veh, seed=rank, keep_host=(rank == 0))
    tm = Timer(torch, dist, use_dist, shard, with_summary=True)
    tm.open_loop = args.open_loop
    tm.run(min(args.warmup, HORIZON), False)
    torch.cuda.synchronize()
    if args.open_loop:
        tm.eager = True
    elif args.eager or args.graph:
        tm.eager = args.eager
    else:
        tm.pick_form()
        if use_dist:                       # every rank runs the same form: rank 0's pick
            flag = torch.tensor([1 if tm.eager else 0], dtype=torch.int32, device=dev if backend == 'nccl' else 'cpu')
            dist.broadcast(flag, 0)
            tm.eager = bool(flag.item())
    r = tm.measure(args.steps, args.warmup, args.repeats)
    dt = median(r['dt'])
    summary = [float(x) for x in combine_summaries(tm.gathered).tolist()]
    per_rank = gather_floats(torch, dist, [r['launch_us'], float(r['launches_timed'])], dev) if use_dist else \
        [[r['launch_us'], float(r['launches_timed'])]]
    job = {'rccl_ranks': dist.get_world_size(), 'backend': dist.get_backend() + (' (RCCL)' if dist.get_backend() == 'nccl' else '')} \
        if use_dist else {'rccl_ranks': 1, 'backend': 'none (single process)'}
    headline_form = 'eager host launches' if tm.eager else 'hipGraph replays of <= %d launches' % HORIZON

    # ---- side measurements (untimed for `value`) ----
    strong = hbm = None
    extra = []
    if not args.no_side and not args.open_loop:
        # side measurements: regions of 20 horizons whatever --steps says (a one-horizon region — the driver's --steps 20 — measures its
        # own launch and wake-up latency: fp16 x 64 read 22.7 us per launch that way against 20.1 in a long region)
        side_steps, side_warm, side_rep = 20 * HORIZON, HORIZON, min(args.repeats, 7)
        _MODELS[n_veh] = model
        m32 = model_for(torch, EnvironmentModel, dev, N_VEH)
        per = STRONG_TOTAL // world
        s = side_config(torch, dist, m32, per, N_VEH, 1000 + rank, side_steps, side_warm, side_rep, use_dist=use_dist)
        if rank == 0:
            strong = dict(s, workload='configs[3]: N_env=%d in total, %d per GPU, N_veh=%d, horizon=%d (strong scaling: the total '
                                      'is fixed as N grows)' % (STRONG_TOTAL, per, N_VEH, HORIZON), n_gpus=world, scaling='strong')
        if world == 1:
            # north_star's ">= 6x at 8 GPUs" seen from ONE GPU: the per-rank shards of configs[3] at N = 2 / 4 / 8 through the
            # headline's protocol (summary kernels + gather per horizon included).  Ranks share nothing on the data path, so the
            # N-GPU time of a step is the time of its slowest shard: t(262144) / t(262144 / N) is what N GPUs give before RCCL's
            # 32-byte all-gather per horizon (asynchronous, off the launch stream) and rank-to-rank jitter.
            # (regions of 20 horizons, as every side measurement: the projection is about the steady state of a long sharded job — a
            # one-horizon region charges its launch / wake-up latency to the smallest shard: 5.2 x instead of 6)
            proj, proj_steps = {}, side_steps
            t1 = side_config(torch, dist, m32, STRONG_TOTAL, N_VEH, 1000, proj_steps, side_warm, side_rep, with_summary=True)
            for n in (2, 4, 8):
                sh = side_config(torch, dist, m32, STRONG_TOTAL // n, N_VEH, 1000, proj_steps, side_warm, side_rep, with_summary=True)
                proj[str(n)] = {'n_env_per_gpu': STRONG_TOTAL // n, 'ms_per_step': sh['ms_per_step'], 'avg_launch_us': sh['avg_launch_us'],
                                'frac': sh['frac'], 'launch_form': sh['launch_form'],
                                'projected_speedup': t1['ms_per_step'] / sh['ms_per_step']}
            strong['projection'] = {'kind': 'ONE-GPU EXTRAPOLATION, NOT a multi-GPU measurement: no RCCL, no second rank ran',
                                    'what': 'per-rank shard of configs[3] at N GPUs timed on this one GPU, same protocol as the headline '
                                            '(episodic summary kernels per horizon inside the timed region; with one rank their gather is a '
                                            'view, no launch), regions of %d steps; ratio = this box\'s own t(262144) / t(262144 / N)' % proj_steps,
                                    'steps_per_region': proj_steps,
                                    'one_gpu_ms_per_step': t1['ms_per_step'], 'one_gpu_frac': t1['frac'], 'by_n_gpus': proj,
                                    'projected_speedup_at_8': proj['8']['projected_speedup'],
                                    'north_star_asks_for': '>= 6x at 8 GPUs, measured (SCALE_rNN.json is the driver\'s measurement when a node is available)',
                                    'box_spread': 'the numerator is this box\'s 262144-env step (53.6-61.2 us across the pool\'s boxes in rounds 4-5, '
                                                  'the working set exceeds the Infinity Cache); the shard\'s step varies by < 2 %',
                                    'not_included': 'RCCL all-gather of 8 floats per rank and horizon (asynchronous, on RCCL\'s own stream), '
                                                    'rank-to-rank jitter'}
        if world == 1:
            lanes = 8
            a = side_config(torch, dist, m32, N_ENV, N_VEH, 0, 100, HORIZON, min(args.repeats, 5), lanes=lanes, forms=('eager',))
            b = side_config(torch, dist, m32, 8 * N_ENV, N_VEH, 7, 100, HORIZON, min(args.repeats, 5))
            hbm = {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                   'achieved': a['achieved_GBs'], 'frac': a['frac'],
                   'exceeds_infinity_cache': 'both: (a) 8 env sets x 65 536 envs stepped round-robin — 7 x 74 MB of other traffic '
                                             'between a line\'s write and its re-read, footprint %.0f MB; (b) one batch of 524 288 envs '
                                             '— 290 MB per obs buffer, footprint %.0f MB; Infinity Cache = 268 MB'
                                             % (a['footprint_MB'], b['footprint_MB']),
                   'lanes8_x_65536': a, 'single_524288': b,
                   # the same 8 independent env sets dealt to four HIP streams: a set's launch may start while another set's
                   # blocks drain (what one generation of co-resident blocks cannot do for itself; wall-clock timed)
                   'lanes8_x_65536_four_streams': side_config(torch, dist, m32, N_ENV, N_VEH, 0, 100, HORIZON, min(args.repeats, 5),
                                                              lanes=lanes, forms=('eager',), streams=4)}
            m16 = model_for(torch, EnvironmentModel, dev, 16)
            extra.append(dict(side_config(torch, dist, m16, 4096, 16, 11, side_steps, side_warm, side_rep),
                              workload='configs[1]: N_env=4096, N_veh=16, horizon=25, fp32 (value: one launch per step)',
                              one_launch_forms=one_launch_forms(torch, m16, 4096, 16, 11)))
            extra.append(dict(side_config(torch, dist, model_for(torch, EnvironmentModel, dev, 64), N_ENV, 64, 12, side_steps, side_warm,
                                          side_rep, f16=True), workload='configs[4]: N_env=65536, N_veh=64, fp16 state / fp32 reward accumulate'))
            # "task left (primary; also report straight, right)" — configs[2]'s shape on the other two tasks, and the
            # reference's own shapes: the native slot counts (UTL:21-23: 8 / 9 / 5 vehicles in the observation) at the same batch size
            other = []
            for task, nv in (('straight', N_VEH), ('right', N_VEH), ('left', 8), ('straight', 9), ('right', 5)):
                r_ = side_config(torch, dist, model_for(torch, EnvironmentModel, dev, nv, task), N_ENV, nv, 13, side_steps, side_warm, side_rep)
                other.append({'task': task, 'n_env_per_gpu': N_ENV, 'n_veh': nv, 'native_slot_count': nv != N_VEH, 'value': r_['value'],
                              'unit': 'env-steps/s', 'ms_per_step': r_['ms_per_step'], 'avg_launch_us': r_['avg_launch_us'],
                              'alg_bytes_per_launch': r_['alg_bytes_per_launch'], 'frac': r_['frac'], 'launch_form': r_['launch_form']})
            extra.append({'workload': 'rollout_out on the other tasks (N_env=65536, N_veh=32) and at the reference\'s native slot counts '
                                      '(left 8, straight 9, right 5), fp32, one launch per step', 'other_tasks_and_native_shapes': other})
            extra.append(env_step_bench(torch, dev, N_ENV))      # the env-side step (endtoend.py), one launch per step
            extra.append(env_step_bench(torch, dev, 4096))
            extra.append(env_step_flows_bench(torch, dev, N_ENV))     # ... over the flow traffic source (60 candidates per env)
            # the drop-in call itself ("H consecutive rollout_out calls"), host-side cost included
            extra.append({'facade_rollout_out': [facade_rollout_bench(torch, EnvironmentModel, dev, N_ENV, N_VEH, 21),
                                                 facade_rollout_bench(torch, EnvironmentModel, dev, 4096, 16, 22)]})

    if rank == 0:
        value = n_env * world * args.steps / dt
        alg = alg_bytes_per_env_step(n_veh) * n_env
        launch_us = r['launch_us']
        traffic, traffic_src = None, None
        if n_env == N_ENV and n_veh == N_VEH and not args.open_loop:      # separate rocprofv3 --pmc passes (scripts/pmc_traffic.sh)
            traffic, traffic_src = pmc_traffic('rollout', 'hbm_bytes_per_launch')
        cfg = 'configs[2]' if (n_env, n_veh) == (N_ENV, N_VEH) else 'custom'
        plan = (C.c_int32 * 4)()
        model.api.debug_rollout_plan(model.handle, n_env, plan)     # what eb_rollout_step launches for this batch on this device
        launch_plan = {'tile_records': (2048, 1024, 256)[plan[0]], 'workgroups': plan[1],
                       'record_loads': 'three in flight per lane, rolling' if plan[2] else 'all up front',
                       'issue_priority': 'by progress (s_setprio)' if plan[3] else 'hardware (oldest first)'}
        kernel = 'eb::rollout_fused_4x8<0, true, 8, float>'   # (task left, slot count divides the record lanes, every record load up front, fp32)
        form = 'closed-loop rollout_out (one kernel launch per step, %s)' % headline_form
        if args.open_loop:
            # one launch per tape; HBM sees the initial and final obs once, actions and outputs every step
            h_eff = min(args.steps, HORIZON)
            alg = (28 * h_eff + 72 + 32 * n_veh) * n_env
            kernel = 'eb::rollout_tape_4x8<0, true, float>'
            form = ('OPEN-LOOP eb_rollout_tape (the whole %d-step action tape in one launch, records and ego state in registers; '
                    'VALU-bound — reported apart from the closed-loop headline)' % h_eff)
        achieved, frac = roofline_of(alg, launch_us)
        ws = shard.footprint_bytes()
        line = {
            'metric': 'env-steps/s (batched rollout) at N_env x N_veh; achieved HBM GB/s vs peak',
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt * 1e3 / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: N_env=%d per GPU, N_veh=%d, horizon=%d, task=%s, mode=training, %s'
                                   % (cfg, n_env, n_veh, min(args.steps, HORIZON), TASK, form),
                       'n_env_per_gpu': n_env, 'n_veh': n_veh, 'horizon': min(args.steps, HORIZON),
                       'parallelism': 'env-shard x%d, all-gather of the 8-float episodic summary per horizon%s'
                                      % (world, '' if use_dist else ' (one process: the gather is a view of the rank\'s own summary, no launch)')},
            'repeats': {'n': len(r['dt']), 'statistic': 'median', 'ms_per_step_min': min(r['dt']) * 1e3 / args.steps,
                        'ms_per_step_median': dt * 1e3 / args.steps, 'ms_per_step_max': max(r['dt']) * 1e3 / args.steps},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': frac, 'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': kernel, 'launch_plan': launch_plan, 'alg_bytes_per_launch': alg,
                         'avg_launch_us': launch_us, 'launches_timed': r['launches_timed'],
                         'avg_launch_us_statistic': 'HIP events around every rollout of the timed regions: mean per launch within a '
                                                    'region, median over the regions (as `value`)',
                         'avg_launch_us_by_region': {'min': min(r['launch_us_by_repeat']), 'max': max(r['launch_us_by_repeat'])},
                         'working_set_MB': ws / 1e6,
                         'residency': ('working set %.0f MB < 268 MB Infinity Cache: this fraction is cache-assisted; see hbm_resident'
                                       % (ws / 1e6)) if ws < MALL_BYTES else 'working set exceeds the Infinity Cache',
                         'hbm_resident': hbm},
            'rccl_ranks': job['rccl_ranks'], 'backend': job['backend'],
            'summary': summary,
            'strong': strong,
            'extra': extra,
        }
        us = [x[0] for x in per_rank]
        agg = sum(alg / (u * 1e-6) / 1e9 for u in us)
        line['roofline']['aggregate'] = {
            'achieved': agg, 'peak': HBM_PEAK_GBS * world, 'unit': 'GB/s', 'frac': agg / (HBM_PEAK_GBS * world),
            'ranks': world, 'avg_launch_us_min': min(us), 'avg_launch_us_max': max(us), 'avg_launch_us_by_rank': us,
            'note': 'sum over ranks of alg_bytes_per_launch / that rank\'s own event-timed launch duration; `achieved` / `frac` '
                    'above are rank 0\'s'}
        if not args.no_cpu_baseline:   # rank 0, after every timed GPU region (the other ranks wait at the final barrier)
            line['cpu_baseline'] = cpu_baseline(shard.inp, shard.obs0_f32.cpu().numpy(), n_veh, budget_s=args.cpu_budget_s)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    tm.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


_MODELS = {}


def model_for(torch, EnvironmentModel, dev, n_veh, task=TASK):
    """EnvironmentModel per slot count and task (the handle fixes both); state dtype is chosen per call by the entry point used"""
    key = n_veh if task == TASK else (task, n_veh)
    if key not in _MODELS:
        _MODELS[key] = EnvironmentModel(task, num_future_data=0, mode='training', n_veh=n_veh, device=dev)
    return _MODELS[key]


if __name__ == '__main__':
    main()
