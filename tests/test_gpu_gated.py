"""-m gpu: eb_rollout_gated — the H-step rollout in one launch with a device-side gate in front of every step — equals
`horizon` per-step launches bit for bit: with every gate opened in advance, with the gates opened one by one by a
producer kernel on another stream (eb_gate_feed: the hand-off a policy kernel in the loop would do), and it gives up
cleanly (status word, no hang) when a gate stays shut."""
import ctypes as C

import numpy as np
import pytest

from env_build_amd import _capi

from env_build_amd.synthetic import assemble_obs, make_rollout_inputs
from tests._helpers import DeviceModel, HostModel, oracle_lib

pytestmark = pytest.mark.gpu


def _initial_obs(host, inp):
    ego = inp['ego']
    trk = host.tracking_error(ego[:, 3], ego[:, 4], ego[:, 5], ego[:, 0], host.n_future, ref_idx=inp['ref_idx'])
    return assemble_obs(ego, trk, inp['veh'])


def _stepwise(host, obs0, inp, H):
    obs, o5s, states = obs0, [], []
    for t in range(H):
        obs, o5, _ = host.rollout_step(obs, inp['actions'][t], inp['ref_idx'])
        o5s.append(o5); states.append(obs)
    return obs, np.stack(o5s), np.stack(states)


def _check(o5_d, o5_h):
    assert np.array_equal(o5_d[:, 0], o5_h[:, 0]) and np.array_equal(o5_d[:, 4], o5_h[:, 4])
    np.testing.assert_allclose(o5_d, o5_h, rtol=1e-6, atol=0)


@pytest.mark.parametrize('task,N,B,tile,nf', [('left', 16, 4096, -1, 0), ('straight', 9, 777, 2, 2), ('right', 32, 1500, 1, 0),
                                            ('left', 32, 3000, 0, 0)])
def test_gated_rollout_with_open_gates_equals_stepwise(task, N, B, tile, nf):
    H = 9
    host, dev = HostModel(oracle_lib(), task, n_veh=N, n_future=nf), DeviceModel(task, n_veh=N, n_future=nf)
    dev.set_tile(tile)
    inp = make_rollout_inputs(task, B, N, H, seed=40 + N, n_future=nf)
    obs0 = _initial_obs(host, inp)
    want_obs, want_o5, want_states = _stepwise(host, obs0, inp, H)
    nb = dev.gated_blocks(B)
    assert nb >= 1
    for publish in (True, False):
        out, o5, steps, done, status = dev.rollout_gated(obs0, inp['actions'], inp['ref_idx'], publish_obs=publish)
        assert status.tolist() == [0, 0] and done.shape == (H, nb, 16) and done.all()
        assert np.array_equal(out, want_obs)
        _check(o5, want_o5)
        if publish:
            assert np.array_equal(steps, want_states)
    # the oracle's twin of the entry point
    out_h, o5_h, steps_h, done_h, st_h = host.rollout_gated(obs0, inp['actions'], inp['ref_idx'])
    assert np.array_equal(out_h, want_obs) and np.array_equal(steps_h, want_states) and done_h.shape == (H, 1, 16) and done_h.all()


@pytest.mark.parametrize('B,N', [(4096, 16), (2048, 32), (1024, 8), (8192, 16), (16384, 32)])
def test_gated_rollout_fed_step_by_step_from_another_stream(B, N):
    """The closed loop without the host: a producer kernel on a second stream releases actions[t] only after every block
    has published step t - 1; the rollout's blocks wait at their gates in between."""
    import torch
    task, H = 'left', 12
    host, dev = HostModel(oracle_lib(), task, n_veh=N), DeviceModel(task, n_veh=N)
    inp = make_rollout_inputs(task, B, N, H, seed=7)
    obs0 = _initial_obs(host, inp)
    want_obs, want_o5, want_states = _stepwise(host, obs0, inp, H)
    t = torch
    d = dev.dev
    ob, staged, ri = dev._in(obs0), dev._in(inp['actions']), dev._in(inp['ref_idx'], np.int32)
    live = t.full_like(staged, float('nan'))                 # nothing usable until the feeder has delivered it
    work, out, out5 = t.empty_like(ob), t.empty_like(ob), t.empty((H, 5, B), device=d)
    steps = t.empty((H,) + tuple(ob.shape), device=d)
    nb = dev.gated_blocks(B)
    ready, done, status = (t.zeros(H, dtype=t.int32, device=d), t.zeros((H, nb, 16), dtype=t.int32, device=d), t.zeros(2, dtype=t.int32, device=d))
    p = lambda x: C.c_void_p(x.data_ptr())
    spin = 1 << 18                                           # ~ a second of polling at most, then both sides give up
    # the producer on the handle's own (high-priority) stream — a hardware queue of its own —, the rollout on torch's
    # (wait_after = 1: the feed is ordered behind torch's stream, where the zero fills above were enqueued — no host sync needed)
    dev.api.gate_feed(dev.h, B, H, nb, p(staged), p(live), p(ready), p(done), p(status), spin, dev.stream, 1, None)
    dev.api.rollout_gated(dev.h, B, H, p(ob), p(live), p(ri), 0, p(work), p(out), p(out5), p(steps), p(ready), p(done), nb, p(status),
                          spin, dev.stream)
    t.cuda.synchronize()
    assert status.cpu().tolist() == [0, 0]
    assert bool(done.cpu().all()) and ready.cpu().tolist() == [1] * H
    assert np.array_equal(live.cpu().numpy(), inp['actions'])
    assert np.array_equal(out.cpu().numpy(), want_obs) and np.array_equal(steps.cpu().numpy(), want_states)
    _check(out5.cpu().numpy(), want_o5)


def test_gated_rollout_gives_up_at_a_shut_gate_and_refuses_oversized_batches():
    task, N, B, H = 'left', 16, 1024, 6
    host, dev = HostModel(oracle_lib(), task, n_veh=N), DeviceModel(task, n_veh=N)
    inp = make_rollout_inputs(task, B, N, H, seed=3)
    obs0 = _initial_obs(host, inp)
    ready = np.ones(H, np.int32)
    ready[3] = 0                                             # nobody will ever open gate 3
    out, o5, steps, done, status = dev.rollout_gated(obs0, inp['actions'], inp['ref_idx'], ready=ready, spin_limit=2000)
    nb = dev.gated_blocks(B)
    assert status[0] == 1 and done.shape == (H, nb, 16) and done[:3].all() and not done[3:].any()
    _, want_o5, want_states = _stepwise(host, obs0, inp, 3)
    assert np.array_equal(steps[:3], want_states)            # what was published before the gate is good
    _check(o5[:3], want_o5)
    # the handle still works afterwards
    o1, _, _ = dev.rollout_step(obs0, inp['actions'][0], inp['ref_idx'])
    assert np.array_equal(o1, want_states[0])
    big = DeviceModel(task, n_veh=32)
    assert big.gated_blocks(16384) == 256                    # 64 envs per 2048-record tile, half of the device's block slots
    assert big.gated_blocks(32768) == 0                      # the other half is the producer's
    assert big.gated_blocks(1 << 20) == 0                    # more blocks than the device holds at once
    with pytest.raises(ValueError):
        big.api.rollout_gated(big.h, 1 << 20, 2, C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), 0, C.c_void_p(16), C.c_void_p(24),
                              C.c_void_p(8), None, C.c_void_p(8), C.c_void_p(8), 1, C.c_void_p(8), 10, None)
    # a block count that is not the grid this handle launches: refused before anything is written (the done records
    # of a larger grid would land past the caller's buffer)
    nb_ok = big.gated_blocks(4096)
    other = [v for v in (0, 1, 2) if (big.set_tile(v), big.gated_blocks(4096))[1] not in (0, nb_ok)]
    assert other                                             # some forced tile shape gives a different (resident) grid
    big.set_tile(other[0])
    assert big.gated_blocks(4096) != nb_ok
    with pytest.raises(ValueError):
        big.api.rollout_gated(big.h, 4096, 2, C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), 0, C.c_void_p(16), C.c_void_p(24),
                              C.c_void_p(8), None, C.c_void_p(8), C.c_void_p(8), nb_ok, C.c_void_p(8), 10, None)
    big.set_tile(-1)
    # the query needs a configured handle (its answer depends on the table size and the slot count)
    bare = big.api.create(task, 32, 0, _capi.MODE_TRAINING)
    with pytest.raises(_capi.EbError):
        big.api.rollout_gated_blocks(bare, 4096, C.byref(C.c_int32()))
    big.api.destroy(bare)
