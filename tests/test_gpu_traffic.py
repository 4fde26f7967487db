"""GPU (-m gpu): the SUMO-free flow traffic source (env_build_amd/traffic.py) — emission schedule of
sumo_files/cross.rou.xml, vTypes, light programme, conflict removal at reset — and that the env built on it still
equals the oracle composition step for step (inactive slots are invisible to the observation and the done test)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from env_build_amd import _capi  # noqa: E402
from tests._helpers import HostModel, oracle_lib  # noqa: E402

pytestmark = pytest.mark.gpu


def test_flow_schedule_types_and_light():
    import torch
    from env_build_amd.endtoend import CrossroadEnd2end
    from env_build_amd.traffic import FLOWS, ROUTES, VTYPES, light_phase
    B = 32
    env = CrossroadEnd2end('left', n_env=B, mode='testing', traffic='flows')
    fl = env._flows
    assert env.n_cand == 60 and fl.M == 60
    # conflict removal at reset (traffic.py:168-192): nothing active in the box ahead of / behind an ego
    ego, cand, act = env._ego.cpu().numpy(), fl.cand.cpu().numpy(), fl.active.cpu().numpy()
    for b in range(B):
        phi = np.deg2rad(ego[b, 5])
        dx, dy = cand[b, :, 0] - ego[b, 3], cand[b, :, 1] - ego[b, 4]
        xe, ye = dx * np.cos(phi) + dy * np.sin(phi), -dx * np.sin(phi) + dy * np.cos(phi)
        reach = ego[b, 0] + 2.4 + fl.lw[:, 0].cpu().numpy() / 2 + 2
        assert not np.any((act[b] != 0) & (xe > -5 + 1e-3) & (xe < reach - 1e-3) & (np.abs(ye) < 3 - 1e-3))
    steps = 600                                                      # 60 s of simulated time
    zero = np.zeros((B, 2), np.float32)
    lights = []
    for t in range(steps):
        env.step(zero)
        lights.append(int(env._v_light[0].item()))
    emitted = fl.emitted.cpu().numpy().astype(np.float64)
    assert np.all(fl.sim_step.cpu().numpy() == steps)
    for k, r in enumerate(ROUTES):
        want = steps * 0.1 * FLOWS[r][0] / 3600.0                    # vehsPerHour spacing
        assert abs(emitted[:, k].mean() - want) < 1.0, (r, emitted[:, k].mean(), want)
        assert emitted[:, k].max() <= want + 1
    # light programme of a.net.xml:145-150 in 'testing' mode: 25 s phase 0, 5 s phase 1, 25 s phase 2, 5 s phase 3
    def phase(n):                                                     # n steps of 0.1 s since the reset
        n %= 600
        return 0 if n < 250 else (1 if n < 300 else (2 if n < 550 else 3))
    assert lights == [phase(t + 1) for t in range(steps)]             # lights[t] is the phase after step t + 1
    assert light_phase(61.0).item() == 0
    # vTypes: length / width / maxSpeed per flow; nobody exceeds its maxSpeed; everybody speeds up to it
    lw, v = fl.lw.cpu().numpy(), fl.cand[:, :, 2].cpu().numpy()
    for j, m in enumerate(fl.slot_modes):
        assert np.allclose(lw[j], VTYPES[FLOWS[m][1]][:2])
        assert np.all(v[:, j] <= VTYPES[FLOWS[m][1]][2] + 1e-6)
    a = fl.active.cpu().numpy() != 0
    assert 1.5 < a.sum(1).mean() / 12 < 5.0                           # a few vehicles per route on the map
    mode = fl.mode().cpu().numpy()
    assert np.all(mode[~a] == _capi.VMODE_EMPTY) and np.all(mode[a] == fl.route_id.cpu().numpy()[None].repeat(B, 0)[a])
    # training mode pins the phase (traffic.py:158-161, 222-223)
    env_r = CrossroadEnd2end('right', n_env=256, mode='training', traffic='flows')
    ph = env_r._v_light.cpu().numpy()
    assert set(np.unique(ph)) <= {0, 2} and 0.3 < (ph == 2).mean() < 0.7
    env_r.step(np.zeros((256, 2), np.float32))
    assert np.array_equal(env_r._v_light.cpu().numpy(), ph)
    torch.cuda.synchronize()


@pytest.mark.parametrize('task', ['left', 'straight', 'right'])
def test_env_on_flow_traffic_equals_oracle_composition(task):
    """obs, reward and done code of every step == the six oracle calls on the same state, with the slot modes the
    traffic source publishes (EB_VMODE_EMPTY for vacant slots)."""
    from env_build_amd.endtoend import CrossroadEnd2end
    from env_build_amd.endtoend_env_utils import VEH_NUM
    B = 48
    env = CrossroadEnd2end(task, n_env=B, mode='testing', traffic='flows')
    host = HostModel(oracle_lib(), task, n_veh=VEH_NUM[task])
    traffic = HostModel(oracle_lib(), task, n_veh=env.n_cand, modes=env.cand_modes)
    rng = np.random.default_rng(3)
    hits = 0
    for t in range(30):
        ego, par = env._ego.cpu().numpy(), env._params.cpu().numpy()
        cand, cmode = env._cand.cpu().numpy(), env._cand_mode.cpu().numpy()
        vl, virt = env._v_light.cpu().numpy(), env._virtual.cpu().numpy()
        lw = env._flows.cand_lw().cpu().numpy()
        ri = env._ref_idx.cpu().numpy()
        obs = env._obs.cpu().numpy()
        raw = rng.uniform(-1, 1, (B, 2)).astype(np.float32)
        o, r, d, info = env.step(raw)
        act = host.action_transform(raw)
        o5, _ = host.compute_rewards(obs, act)
        ego2, par2 = host.env_ego_step(ego, act)
        cand2 = traffic.veh_predict(cand.reshape(B, -1)).reshape(cand.shape)
        obs2 = host.get_obs(ego2, cand2, cmode, vl, ref_idx=ri, virtual=virt)
        code = host.judge_done(ego2, par2, obs2, cand2, cmode, lw, vl)       # the vTypes' (l, w), TRF:263-295
        assert np.array_equal(o.numpy(), obs2), 'obs, step %d' % t
        assert np.array_equal(r.numpy(), o5[0]), 'reward, step %d' % t
        assert np.array_equal(env.done_code.cpu().numpy(), code), 'done code, step %d' % t
        hits += int((code != 0).sum())
        if (code != 0).any():
            env.reset(mask=code != 0)
    assert hits > 0


def test_traffic_respawn_equals_oracle_and_depends_on_its_key_only():
    """eb_traffic_respawn: bit-exact against the oracle (integer hash + three fp32 ops), and a function of
    (seed, counter, env, slot) only — the same key gives the same draw in any batch."""
    from tests._helpers import DeviceModel
    import ctypes as C
    rng = np.random.default_rng(8)
    B, M = 333, 16
    host, dev = HostModel(oracle_lib(), 'left', n_veh=M), DeviceModel('left', n_veh=M)
    cand = (rng.uniform(-90, 90, (B, M, 4))).astype(np.float32)
    entry = rng.uniform(-60, 60, (M, 5)).astype(np.float32)
    outs = []
    for mdl in (host, dev):
        c, en = mdl._in(cand.copy()), mdl._in(entry)
        flag = mdl._out((B, M), np.uint8)
        mdl.api.traffic_respawn(mdl.h, B, M, mdl._ptr(c), mdl._ptr(en), C.c_float(65.0), C.c_float(60.0), C.c_float(8.0),
                                C.c_uint64(12345678901234567), C.c_uint64(77), None, mdl._ptr(flag), None, C.c_float(0.), mdl.stream)
        outs.append((mdl._ret(c), mdl._ret(flag)))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    got, flag = outs[1]
    gone = (np.abs(cand[:, :, 0]) > 65) | (np.abs(cand[:, :, 1]) > 65)
    assert np.array_equal(flag.astype(bool), gone) and 0.2 < gone.mean() < 0.8
    assert np.array_equal(got[~gone], cand[~gone])
    along = (got[gone][:, 0] - np.broadcast_to(entry[:, 0], (B, M))[gone]) / np.broadcast_to(entry[:, 3], (B, M))[gone]
    assert np.all(got[gone][:, 2] >= 0) and np.all(got[gone][:, 2] < 8) and np.all(along > -1e-3) and np.all(along < 60.001)
    # the same (env, slot) rows inside a smaller batch draw the same values
    c2, en2 = dev._in(cand[:100].copy()), dev._in(entry)
    dev.api.traffic_respawn(dev.h, 100, M, dev._ptr(c2), dev._ptr(en2), C.c_float(65.0), C.c_float(60.0),
                            C.c_float(8.0), C.c_uint64(12345678901234567), C.c_uint64(77), None, None, None, C.c_float(0.), dev.stream)
    assert np.array_equal(dev._ret(c2), got[:100])
    u = got[gone][:, 2] / 8.0
    assert abs(u.mean() - 0.5) < 0.03                                   # roughly uniform draws


def test_flow_step_kernel_equals_oracle():
    """eb_traffic_flow_step: device == oracle bit for bit over 300 steps of a free-running traffic state (exits,
    acceleration, emissions, slot modes, clock, light), each side advancing its slots with its own eb_veh_predict."""
    import ctypes as C
    from tests._helpers import DeviceModel
    from env_build_amd.traffic import ACCEL, EXIT_RANGE, FLOWS, LANE_START, ROUTES, VTYPES, approach_lane
    K, B = 5, 40
    M = 12 * K
    slot_modes = [r for r in ROUTES for _ in range(K)]
    host, dev = HostModel(oracle_lib(), 'left', n_veh=M, modes=slot_modes), DeviceModel('left', n_veh=M, modes=slot_modes)
    lane = np.array([list(approach_lane(m)[0]) + list(approach_lane(m)[1]) for m in slot_modes], np.float32)
    period = np.array([3600.0 / FLOWS[r][0] for r in ROUTES], np.float32)
    vmax = np.array([VTYPES[FLOWS[m][1]][2] for m in slot_modes], np.float32)
    rng = np.random.default_rng(5)
    state0 = dict(cand=np.zeros((B, M, 4), np.float32), active=np.zeros((B, M), np.uint8),
                  timer=(rng.random((B, 12)) * period).astype(np.float32), emitted=np.zeros((B, 12), np.int32),
                  sim_step=np.zeros((B,), np.int32))
    res = []
    for mdl in (host, dev):
        st = {k: mdl._in(v.copy(), v.dtype) if k in ('cand', 'timer') else None for k, v in state0.items()}
        if mdl is host:
            st = {k: v.copy() for k, v in state0.items()}
            ln, pe, vm = lane, period, vmax
            mode, light = np.zeros((B, M), np.uint8), np.zeros((B,), np.uint8)
            ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        else:
            t = mdl.torch
            st = {k: t.from_numpy(v.copy()).to(mdl.dev) for k, v in state0.items()}
            ln, pe, vm = (t.from_numpy(x).to(mdl.dev) for x in (lane, period, vmax))
            mode, light = t.zeros((B, M), dtype=t.uint8, device=mdl.dev), t.zeros((B,), dtype=t.uint8, device=mdl.dev)
            ptr = lambda a: C.c_void_p(a.data_ptr())
        trace = []
        for step in range(300):
            flat = st['cand'].reshape(B, 4 * M)
            mdl.api.veh_predict(mdl.h, B, ptr(flat), ptr(flat), mdl.stream)          # in place, as eb_env_step does
            mdl.api.traffic_flow_step(mdl.h, B, K, ptr(st['cand']), ptr(st['active']), ptr(st['timer']), ptr(st['emitted']),
                                      ptr(st['sim_step']), ptr(ln), ptr(pe), ptr(vm), C.c_float(0.1), C.c_float(EXIT_RANGE),
                                      C.c_float(ACCEL), C.c_float(LANE_START - 25.0), 1, C.c_uint64(99), C.c_uint64(step + 1),
                                      ptr(mode), ptr(light), mdl.stream)
            if step % 50 == 49:
                get = (lambda a: a.copy()) if mdl is host else (lambda a: (mdl.torch.cuda.synchronize(), a.cpu().numpy())[1])
                trace.append({k: get(v) for k, v in dict(st, mode=mode, light=light).items()})
        res.append(trace)
    for a, b in zip(*res):
        on = a['active'] != 0
        assert np.array_equal(a['active'], b['active']) and np.array_equal(a['mode'], b['mode'])
        assert np.array_equal(a['cand'][on], b['cand'][on])
        for k in ('timer', 'emitted', 'sim_step', 'light'):
            assert np.array_equal(a[k], b[k]), k
    last = res[1][-1]
    assert last['emitted'].min() >= 4 and (last['active'] != 0).sum() > 12 * B and set(np.unique(last['light'])) <= {0, 1, 2, 3}


def test_single_env_predicates_agree_with_the_done_code():
    """n_env == 1: the reference's predicate methods (E2E:223-256) on the published host state reproduce the
    priority chain of the done code the kernel returned (collision aside, which only the kernel sees)."""
    from env_build_amd.endtoend import CrossroadEnd2end
    rng = np.random.default_rng(12)
    seen = set()
    for task in ('left', 'straight', 'right'):
        env = CrossroadEnd2end(task, n_env=1, mode='testing', traffic='flows')
        env.reset()
        for t in range(250):
            a = rng.uniform(-1, 1, 2).astype(np.float32) if t % 40 < 30 else np.array([rng.choice([-1., 1.]), 1.], np.float32)
            obs, r, done, info = env.step(a)
            if env.done_type != 'collision':
                chain = ('break_road_constrain' if env._break_road_constrain() else
                         'deviate_too_much' if env._deviate_too_much() else
                         'break_stability' if env._break_stability() else
                         'break_red_light' if env._break_red_light() else
                         'good_done' if env._is_achieve_goal() else 'not_done_yet')
                assert chain == env.done_type, (task, t, chain, env.done_type)
            seen.add(env.done_type)
            if done:
                env.reset()
    assert len(seen) >= 3


def test_pool_reset_keeps_clear_of_the_ego_and_equals_oracle():
    """eb_traffic_respawn(ego=...): TRF:168-192's conflict rule for the pool, GPU == oracle bit for bit"""
    from tests._env_step_check import respawn_conflict_case
    from tests._helpers import DeviceModel, HostModel, oracle_lib
    got = respawn_conflict_case(lambda t, **kw: DeviceModel(t, **kw))
    want = respawn_conflict_case(lambda t, **kw: HostModel(oracle_lib(), t, **kw))
    assert np.array_equal(got, want)
