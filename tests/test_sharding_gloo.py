"""world_size-2 gloo test (CPU) of the multi-GPU path: contiguous env shards, no data-path collective, one
all-gather of the 8-float episodic summary.  The compute leg on CPU is the oracle (test infrastructure);
the product code under test is env_build_amd/sharding.py — the same functions bench.py runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from env_build_amd.sharding import combine_summaries, gather_summaries, gather_summaries_async, shard_range  # noqa: E402

TASK, B, N, H = 'left', 1001, 8, 5


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _shard_summary(lo, hi):
    from env_build_amd.synthetic import assemble_obs, make_rollout_inputs
    from tests._helpers import HostModel, oracle_lib
    host = HostModel(oracle_lib(), TASK, n_veh=N)
    inp = make_rollout_inputs(TASK, B, N, H, seed=9)
    ego, ref = inp['ego'][lo:hi], inp['ref_idx'][lo:hi]
    trk = host.tracking_error(ego[:, 3], ego[:, 4], ego[:, 5], ego[:, 0], 0, ref_idx=ref)
    obs0 = assemble_obs(ego, trk, inp['veh'][lo:hi])
    out, o5 = host.rollout_tape(obs0, inp['actions'][:, lo:hi], ref)
    return host.episode_summary(o5, out), out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        lo, hi = shard_range(B, rank, world)
        s8, out = _shard_summary(lo, hi)
        all8 = gather_summaries(torch.from_numpy(s8))
        # the overlapped form bench.py uses: two gathers in flight, collected oldest first
        pend = [gather_summaries_async(torch.from_numpy(s8)), gather_summaries_async(torch.from_numpy(s8 * 2))]
        assert torch.equal(pend[0].result(), all8) and torch.equal(pend[1].result(), all8 * 2)
        total = combine_summaries(all8)
        # bench.py's per-rank figures at N > 1 (launch durations for roofline.aggregate): every rank ends with all of them
        import bench
        per_rank = bench.gather_floats(torch, dist, [10.0 + rank, 400.0], torch.device('cpu'))
        assert per_rank == [[10.0, 400.0], [11.0, 400.0]] and dist.get_world_size() == world and dist.get_backend() == 'gloo'
        q.put((rank, lo, hi, all8.numpy(), total.numpy(), out[:2].copy()))
    finally:
        dist.destroy_process_group()


def test_shard_ranges_cover_the_batch_once():
    for n, w in ((1001, 2), (65536, 8), (5, 8), (0, 3), (262144, 8)):
        r = [shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_single_process_gather_is_identity():
    s = torch.arange(8, dtype=torch.float32)
    assert torch.equal(gather_summaries(s), s.reshape(1, 8))
    assert torch.equal(gather_summaries_async(s).result(), s.reshape(1, 8))
    assert torch.equal(combine_summaries(s.reshape(1, 8)), s)


def test_two_rank_gloo_rollout_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref8, ref_out = _shard_summary(0, B)                       # the whole batch in one process
    (r0, lo0, hi0, all0, tot0, out0), (r1, lo1, hi1, all1, tot1, out1) = got
    assert (lo0, hi0, lo1, hi1) == (0, 501, 501, 1001)
    assert np.array_equal(all0, all1) and np.array_equal(tot0, tot1)        # every rank holds the same gather
    assert all0.shape == (2, 8) and all0[0, 6] == 501 and all0[1, 6] == 500
    np.testing.assert_allclose(tot0[:3], ref8[:3], rtol=1e-6)               # float64 sums, different grouping
    np.testing.assert_allclose(tot0[4], ref8[4], rtol=1e-6)
    assert tot0[3] == ref8[3] and tot0[5] == ref8[5] and tot0[6] == B and tot0[7] == H
    # no data-path exchange: a shard's rows are exactly the rows of the unsharded run
    assert np.array_equal(out0, ref_out[0:2]) and np.array_equal(out1, ref_out[501:503])
