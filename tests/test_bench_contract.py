"""bench.py's protocol: segmenting of the timed steps (CPU) and one small run under the driver's own flags (GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_segments_cover_exactly_the_requested_steps():
    sys.path.insert(0, ROOT)
    import bench
    for k in (1, 5, 20, 25, 26, 49, 50, 2000, 2013):
        segs = bench.Timer.segments(k)
        assert sum(segs) == k and all(1 <= s <= bench.HORIZON for s in segs)
        assert segs[:-1] == [bench.HORIZON] * (len(segs) - 1)          # only the last rollout may be shorter
    assert bench.median([3.0, 1.0, 2.0]) == 2.0 and bench.median([4.0, 1.0, 2.0, 3.0]) == 2.5
    assert bench.alg_bytes_per_env_step(32) == 1128 and bench.alg_bytes_per_env_step(64, f16=True) == 1092


def test_self_launch_command_is_the_contract_form():
    """`python bench.py --gpus N` without a launcher re-executes itself as the driver would have started it"""
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.self_launch_argv(['--gpus', '4', '--steps', '20', '--warmup', '5'], 4, port=29511)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert cmd[3:10] == ['--nnodes=1', '--nproc-per-node', '4', '--master-addr', '127.0.0.1', '--master-port', '29511']
    assert cmd[10] == os.path.join(ROOT, 'bench.py') and cmd[11:] == ['--gpus', '4', '--steps', '20', '--warmup', '5']
    free = bench.self_launch_argv([], 2)
    assert 1024 < int(free[9]) < 65536


@pytest.mark.gpu
def test_driver_invocation_times_events_summary_and_repeats():
    """`--steps 20 --warmup 5` (fewer steps than one horizon): the rollout is planned with horizon 20, so the event
    pairs, the summary kernels and the all-gather all run, and the label names the launch form that ran."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '20', '--warmup', '5',
                          '--n-env', '8192', '--no-cpu-baseline', '--no-side', '--repeats', '5'],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['steps'] == 20 and line['warmup'] == 5 and line['n_gpus'] == 1
    assert line['roofline']['launches_timed'] == 5 * 20
    assert line['repeats']['n'] == 5 and line['repeats']['ms_per_step_min'] <= line['ms_per_step'] <= line['repeats']['ms_per_step_max']
    s = line['summary']
    assert s[6] == 8192 and s[7] == 20 and s[0] != 0.0                  # n_env, horizon, sum of rewards
    assert ('eager' in line['config']['workload']) != ('hipGraph' in line['config']['workload'])
    assert abs(line['value'] - 8192 * 20 / (line['ms_per_step'] * 1e-3 * 20)) / line['value'] < 1e-9
    # the per-launch figure: median over the regions, and no region pays for building its launch form (hipGraph / argument tuples of
    # a 20-step rollout are made before the clock starts — they used to land in the first region when the warm-up was shorter)
    by = line['roofline']['avg_launch_us_by_region']
    assert by['min'] <= line['roofline']['avg_launch_us'] <= by['max'] and by['max'] < 3.0 * by['min'], by


@pytest.mark.gpu
def test_one_launch_forms_of_configs1_run_and_report():
    """the configs[1] side measurement: gated with open gates, gated fed by the second stream, open-loop tape"""
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from env_build_amd.dynamics_and_models import EnvironmentModel
    model = EnvironmentModel(bench.TASK, num_future_data=0, mode='training', n_veh=16, device=torch.device('cuda', 0))
    r = bench.one_launch_forms(torch, model, 4096, 16, 11, reps=3)
    assert r['horizon'] == bench.HORIZON and r['gated_blocks'] >= 1
    for k in ('gated_open_gates', 'gated_fed_by_second_stream', 'gated_fed_ordered_behind_the_caller_stream', 'open_loop_tape'):
        assert r[k]['value'] > 0 and r[k]['us_per_step'] > 0 and r[k]['unit'] == 'env-steps/s'


@pytest.mark.gpu
def test_two_rank_control_flow_over_gloo_on_one_gpu():
    """RCCL refuses two ranks on one GPU, so on a 1-GPU box the N > 1 path of bench.py runs with EB_BENCH_BACKEND=gloo
    and both ranks pinned to device 0: sharded seeds, barriers, max over ranks, the summary all-gather and its fold, the
    strong-scaling split — everything but the RCCL transport.  (The numbers of such a run mean nothing.)
    Started as PLAIN `python bench.py --gpus 2`: bench.py launches its own two ranks (round 3)."""
    env = dict(os.environ, EB_BENCH_DEVICE='0', EB_BENCH_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20',
                          '--warmup', '5', '--n-env', '8192', '--cpu-budget-s', '2', '--repeats', '3'],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                              # rank 0 alone prints
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and 'x2' in line['config']['parallelism']
    assert line['rccl_ranks'] == 2 and line['backend'] == 'gloo'        # what torch.distributed saw
    assert line['summary'][6] == 2 * 8192 and line['summary'][7] == 20  # both shards' envs in the gathered summary
    assert line['strong']['n_gpus'] == 2 and line['strong']['n_env_per_gpu'] == 262144 // 2 and line['strong']['scaling'] == 'strong'
    assert line['roofline']['hbm_resident'] is None and line['extra'] == []
    agg = line['roofline']['aggregate']
    assert agg['ranks'] == 2 and len(agg['avg_launch_us_by_rank']) == 2 and agg['peak'] == 2 * 8000.0
    assert agg['avg_launch_us_min'] <= line['roofline']['avg_launch_us'] <= agg['avg_launch_us_max']
    cb = line['cpu_baseline']                                           # the CPU leg runs at N > 1 too, on rank 0
    assert cb is not None and cb['value'] > 0 and cb['kind'] == 'port' and cb['cores'] >= 1
