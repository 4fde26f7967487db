"""-m gpu: the round-2 entry points — device-side reset of the env and of both traffic sources, the exit-relative frames
of the 12-ego scene, the small ReferencePath helpers — against the CPU oracle through the same C-ABI (bit for bit), the
reference-generated fixtures G6X / G11 on the GPU, and the façade paths that use them."""
import ctypes as C

import numpy as np
import pytest

from env_build_amd import _capi
from env_build_amd.endtoend_env_utils import VEH_NUM, VEHICLE_MODE_LIST
from tests._helpers import DeviceModel, HostModel, close, golden, oracle_lib

pytestmark = pytest.mark.gpu
TASKS = ('left', 'straight', 'right')


def _pair(task, **kw):
    return HostModel(oracle_lib(), task, **kw), DeviceModel(task, **kw)


# ---- eb_env_reset ----------------------------------------------------------------------------------
@pytest.mark.parametrize('task', TASKS)
def test_env_reset_equals_oracle_and_follows_the_reference_rules(task):
    host, dev = _pair(task, mode='training')
    B = 5000
    rng = np.random.default_rng(3)
    ego0 = rng.normal(0, 1, (B, 6)).astype(np.float32)
    par0 = rng.normal(0, 1, (B, 4)).astype(np.float32)
    ref0 = np.full((B,), 9, np.int32)
    mask = (rng.random(B) < 0.6).astype(np.uint8)
    for training in (1, 0):
        for counter in (1, 2):
            a = host.env_reset(B, 0x1234567, counter, training, ego0, par0, ref0, mask)
            b = dev.env_reset(B, 0x1234567, counter, training, ego0, par0, ref0, mask)
            for x, y in zip(a, b):
                assert np.array_equal(x, y)
            ego, par, ref, virt, done = b
            on, off = mask != 0, mask == 0
            # untouched rows
            assert np.array_equal(ego[off], ego0[off]) and np.array_equal(par[off], par0[off]) and (ref[off] == 9).all()
            assert (virt[off] == 7).all() and (done[off] == 7).all()
            # E2E:472-499: start speed in [0, 8), at rest otherwise; pose = a point of the env's own path inside the span
            assert (ego[on, 0] >= 0).all() and (ego[on, 0] < 8).all() and (ego[on, 1:3] == 0).all()
            assert np.array_equal(par[on], np.tile(np.float32([0, 0, 0.8, 0.8]), (on.sum(), 1)))     # E2E:110-113
            assert set(np.unique(ref[on])) == {0, 1, 2} and (done[on] == 0).all()
            span = {'left': 1400, 'straight': 1700, 'right': 920}[task]
            for k in range(3):
                rows = on & (ref == k)
                px, py, pphi = host.paths[k]
                lo, hi = 700, min(700 + span, len(px)) - 1
                idx = np.array([np.flatnonzero((px == x) & (py == y))[0] for x, y in ego[rows][:200, 3:5]])
                assert idx.min() >= lo and idx.max() <= hi
                assert np.array_equal(pphi[idx], ego[rows][:200, 5])
            frac = virt[on].mean()
            assert (0.05 < frac < 0.15) if training else frac == 0      # E2E:120-126: U > 0.9 in training mode only
    # a draw depends on (seed, counter, env) only: the same envs inside a smaller batch give the same state
    small = dev.env_reset(100, 0x1234567, 2, 0, ego0[:100], par0[:100], ref0[:100], mask[:100])
    assert np.array_equal(small[0], b[0][:100]) and np.array_equal(small[2], b[2][:100])
    other = dev.env_reset(B, 0x1234567, 3, 0, ego0, par0, ref0, mask)
    assert not np.array_equal(other[0][mask != 0], b[0][mask != 0])


def test_env_facade_batched_reset_is_one_kernel_state_and_keeps_the_old_flag_for_its_observation():
    """CrossroadEnd2end.reset(mask=...) for a batch: state == eb_env_reset of the oracle under the env's own key; the reset
    observation is built with the PREVIOUS virtual-red-light flags (E2E:116 precedes E2E:120-126)."""
    from env_build_amd.endtoend import CrossroadEnd2end
    B = 400
    env = CrossroadEnd2end('left', n_env=B, mode='training')
    env.seed(11)
    env.reset()
    host = HostModel(oracle_lib(), 'left', mode='training')
    for it in range(3):
        mask = (np.random.default_rng(it).random(B) < 0.5)
        ego0, par0, ref0 = env._ego.cpu().numpy(), env._params.cpu().numpy(), env._ref_idx.cpu().numpy()
        virt0, cand0 = env._virtual.cpu().numpy(), env._cand.cpu().numpy()
        obs_prev = env._obs.cpu().numpy()
        counter = env._reset_counter + 1
        obs = env.reset(mask=mask)
        want = host.env_reset(B, env._respawn_seed ^ env._RESET_SALT, counter, 1, ego0, par0, ref0, mask.astype(np.uint8))
        assert np.array_equal(env._ego.cpu().numpy(), want[0]) and np.array_equal(env._params.cpu().numpy(), want[1])
        assert np.array_equal(env._ref_idx.cpu().numpy(), want[2])
        virt_new = np.where(mask, want[3], virt0)
        assert np.array_equal(env._virtual.cpu().numpy(), virt_new)
        cand = env._cand.cpu().numpy()
        assert np.array_equal(cand[~mask], cand0[~mask]) and not np.array_equal(cand[mask], cand0[mask])
        o_want = host.get_obs(want[0], cand, env._cand_mode.cpu().numpy(), env._v_light.cpu().numpy(), ref_idx=want[2],
                              virtual=virt0)                                   # the OLD flags
        assert np.array_equal(obs.numpy()[mask], o_want[mask])                 # the reset envs' rows
        assert np.array_equal(obs.numpy()[~mask], obs_prev[~mask])            # the others keep the observation they had (row_mask)
        assert (env.done_type.numpy()[mask] == 0).all()
        env.step(np.zeros((B, 2), np.float32))


@pytest.mark.parametrize('task', TASKS)
def test_reset_pool_composite_equals_the_single_calls_and_the_oracle(task):
    """eb_env_reset_pool (the masked reset as one call) on both libraries"""
    from tests._env_step_check import reset_pool_case
    got = reset_pool_case(lambda t, **kw: DeviceModel(t, **kw), task)
    want = reset_pool_case(lambda t, **kw: HostModel(oracle_lib(), t, **kw), task)
    names = ('ego', 'params', 'ref_idx', 'virtual', 'v_light', 'done_code', 'cand', 'obs')
    for a, b in zip(got, want):
        for k, (x, y) in enumerate(zip(a, b)):
            assert np.array_equal(x, y), (names[k], int((x != y).sum()))


@pytest.mark.parametrize('task,B,M,NV,tile', [('left', 1000, 16, None, 0), ('straight', 333, 16, None, 1), ('right', 130, 33, None, 2),
                                              ('left', 77, 60, 16, None), ('straight', 4097, 5, 32, None)])
def test_reset_pool_one_launch_every_tile_shape(task, B, M, NV, tile):
    """the one-launch form (env_reset_pool_kernel) at 64- / 32- / 16-env tiles, ragged last tiles, more candidates than a chunk
    group, non-native slot lists: the single calls on the same library (inside reset_pool_case) and the oracle"""
    from tests._env_step_check import reset_pool_case
    got = reset_pool_case(lambda t, **kw: DeviceModel(t, **kw), task, B=B, M=M, seed=5 + B, tile=tile, NV=NV)
    want = reset_pool_case(lambda t, **kw: HostModel(oracle_lib(), t, **kw), task, B=B, M=M, seed=5 + B, NV=NV)
    for a, b in zip(got, want):
        for k, (x, y) in enumerate(zip(a, b)):
            assert np.array_equal(x, y), (k, int((x != y).sum()))


def test_reset_pool_unaligned_candidates_take_the_four_launches():
    """a candidate buffer that is not 16-byte aligned cannot use the tile kernels: same results through the separate launches"""
    import torch
    from env_build_amd.endtoend import _lane_entry
    from env_build_amd import _capi
    import ctypes as C
    task, B, M = 'left', 300, 12
    dev = DeviceModel(task, mode='training')
    native = VEHICLE_MODE_LIST[task]
    modes = [native[i % len(native)] for i in range(M)]
    tr = DeviceModel(task, n_veh=M, modes=modes)
    entry = np.array([list(_lane_entry(m)[:3]) + list(_lane_entry(m)[3]) for m in modes], np.float32)
    rng = np.random.default_rng(3)
    cand = rng.normal(scale=30, size=(B, M, 4)).astype(np.float32)
    cmode = np.tile(np.array([_capi.VMODE_ID[m] for m in modes], np.uint8), (B, 1))
    ego = rng.normal(size=(B, 6)).astype(np.float32); params = rng.normal(size=(B, 4)).astype(np.float32)
    ref = np.zeros(B, np.int32); virtual = (rng.random(B) < 0.5).astype(np.uint8); v_light = rng.integers(0, 4, B).astype(np.uint8)
    obs = rng.normal(size=(B, 9 + 4 * len(native))).astype(np.float32)
    mask = (rng.random(B) < 0.4).astype(np.uint8)
    pool = dict(entry=entry, span=60.0, v_max=8.0, seed=7, counter=3, edge_span=5.0)
    want = dev.env_reset_pool(tr, 11, 2, 1, ego, params, ref, virtual, v_light, cand, cmode, obs, pool, mask=mask)
    # the same call with the candidates at a 4-byte offset
    d = 'cuda:0'
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    raw = torch.zeros(B * M * 4 + 1, dtype=torch.float32, device=d)
    cand_u = raw[1:].view(B, M, 4); cand_u.copy_(t(cand))
    assert cand_u.data_ptr() % 16 == 4
    te, tp, tr_, tv, tl, to = t(ego), t(params), t(ref), t(virtual), t(v_light), t(obs)
    dc = torch.full((B,), 7, dtype=torch.uint8, device=d)
    tm, tcm, ten = t(mask), t(cmode), t(entry)
    rule = _capi.EbRespawn(ten.data_ptr(), 0.0, 60.0, 8.0, 7, 3, 5.0)
    P = lambda x: C.c_void_p(x.data_ptr())
    dev.api.env_reset_pool(dev.h, tr.h, B, P(tm), C.c_uint64(11), C.c_uint64(2), 1, P(te), P(tp), P(tr_), P(tv), P(tl), P(dc), None, M,
                           P(cand_u), P(tcm), C.byref(rule), P(to), None, None, None)
    torch.cuda.synchronize()
    got = [te, tp, tr_, tv, tl, dc, cand_u, to]
    for k, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g.cpu().numpy(), w), k


# ---- traffic: pool reset (masked, unconditional re-entry) and the flow source's reset ---------------------
def test_traffic_respawn_mask_and_forced_reentry_equal_oracle():
    rng = np.random.default_rng(8)
    B, M = 257, 12
    host, dev = _pair('left', n_veh=M)
    cand = rng.uniform(-90, 90, (B, M, 4)).astype(np.float32)
    entry = rng.uniform(-60, 60, (M, 5)).astype(np.float32)
    mask = (rng.random(B) < 0.5).astype(np.uint8)
    for limit in (65.0, -1.0):
        outs = []
        for mdl in (host, dev):
            c, en, mk = mdl._in(cand.copy()), mdl._in(entry), mdl._in(mask, np.uint8)
            flag = mdl._out((B, M), np.uint8)
            mdl.api.traffic_respawn(mdl.h, B, M, mdl._ptr(c), mdl._ptr(en), C.c_float(limit), C.c_float(60.0), C.c_float(8.0),
                                    C.c_uint64(987654321), C.c_uint64(5), mdl._ptr(mk), mdl._ptr(flag), None, C.c_float(0.0), mdl.stream)
            outs.append((mdl._ret(c), mdl._ret(flag)))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        got, flag = outs[1]
        gone = (np.abs(cand[:, :, 0]) > limit) | (np.abs(cand[:, :, 1]) > limit) if limit >= 0 else np.ones((B, M), bool)
        gone &= mask[:, None] != 0
        assert np.array_equal(flag != 0, gone) and np.array_equal(got[~gone], cand[~gone])


@pytest.mark.parametrize('task', TASKS)
def test_traffic_flow_reset_equals_oracle(task):
    from env_build_amd.traffic import FLOWS, LANE_START, ROUTES, VTYPES, approach_lane
    K, B = 5, 700
    M = 12 * K
    slot_modes = [r for r in ROUTES for _ in range(K)]
    host, dev = _pair(task, n_veh=M, modes=slot_modes)
    lane = np.array([list(approach_lane(m)[0]) + list(approach_lane(m)[1]) for m in slot_modes], np.float32)
    period = np.array([3600.0 / FLOWS[r][0] for r in ROUTES], np.float32)
    vmax = np.array([VTYPES[FLOWS[m][1]][2] for m in slot_modes], np.float32)
    vlen = np.array([VTYPES[FLOWS[m][1]][0] for m in slot_modes], np.float32)
    rng = np.random.default_rng(5)
    env_h = HostModel(oracle_lib(), task, mode='training')
    ego = env_h.env_reset(B, 77, 1, 1, np.zeros((B, 6), np.float32), np.zeros((B, 4), np.float32), np.zeros(B, np.int32))[0]
    mask = (rng.random(B) < 0.7).astype(np.uint8)
    state0 = dict(cand=rng.uniform(-50, 50, (B, M, 4)).astype(np.float32), active=(rng.random((B, M)) < 0.5).astype(np.uint8),
                  timer=rng.random((B, 12)).astype(np.float32), emitted=rng.integers(0, 9, (B, 12)).astype(np.int32),
                  sim_step=rng.integers(0, 99, B).astype(np.int32), phase0=np.full(B, 9, np.uint8),
                  mode=np.full((B, M), 77, np.uint8), light=np.full(B, 9, np.uint8))
    res = []
    for mdl in (host, dev):
        st = {k: mdl._in(v.copy(), v.dtype) for k, v in state0.items()}
        p = mdl._ptr
        # (inputs are held in variables: a device tensor created inline would be freed before the kernel reads it)
        mk, eg, ln, pe, vm, vl = (mdl._in(mask, np.uint8), mdl._in(ego), mdl._in(lane), mdl._in(period), mdl._in(vmax),
                                  mdl._in(vlen))
        mdl.api.traffic_flow_reset(mdl.h, B, K, p(mk), p(eg), p(st['cand']), p(st['active']),
                                   p(st['timer']), p(st['emitted']), p(st['sim_step']), p(st['phase0']), p(ln),
                                   p(pe), p(vm), p(vl), C.c_float(LANE_START - 25.0),
                                   1 if task == 'right' else 0, 1, C.c_uint64(4242), C.c_uint64(3), p(st['mode']), p(st['light']),
                                   mdl.stream)
        res.append({k: mdl._ret(v) for k, v in st.items()})
        del mk, eg, ln, pe, vm, vl
    a, b = res
    on = b['active'] != 0
    for k in ('active', 'mode', 'timer', 'emitted', 'sim_step', 'phase0', 'light'):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a['cand'][on], b['cand'][on])
    m = mask != 0
    # untouched envs keep everything; chosen envs: clock and counters restart, every active slot carries its route id
    for k in ('cand', 'active', 'timer', 'emitted', 'sim_step', 'phase0', 'mode', 'light'):
        assert np.array_equal(b[k][~m], state0[k][~m]), k
    assert (b['sim_step'][m] == 0).all() and (b['emitted'][m] == 0).all()
    route = np.repeat(np.arange(12, dtype=np.uint8), K)[None].repeat(B, 0)
    assert np.array_equal(b['mode'][m], np.where(on[m], route[m], _capi.VMODE_EMPTY))
    assert (b['timer'][m] >= 0).all() and (b['timer'][m] < period[None]).all()
    assert set(np.unique(b['phase0'][m])) == ({0, 2} if task == 'right' else {0})        # TRF:158-161
    assert np.array_equal(b['light'][m], b['phase0'][m])                                   # training pins the phase, TRF:222-223
    assert 1.0 < on[m].sum(1).mean() / 12 < 4.0
    # placed vehicles sit on their approach lane with a speed below their vType's maxSpeed
    along = (b['cand'][..., 0] - lane[None, :, 0]) * lane[None, :, 3] + (b['cand'][..., 1] - lane[None, :, 1]) * lane[None, :, 4]
    sel = on & m[:, None]
    assert (along[sel] >= 0).all() and (along[sel] < 75.0 + 1e-3).all() and (b['cand'][..., 2][sel] < vmax[None].repeat(B, 0)[sel]).all()
    # and none of them conflicts with the ego (the float64 predicate of TRF:168-192, off-threshold)
    for e in np.flatnonzero(m)[:150]:
        phi = np.deg2rad(float(ego[e, 5]))
        dx, dy = b['cand'][e, :, 0].astype(np.float64) - ego[e, 3], b['cand'][e, :, 1].astype(np.float64) - ego[e, 4]
        xe, ye = dx * np.cos(phi) + dy * np.sin(phi), -dx * np.sin(phi) + dy * np.cos(phi)
        reach = ego[e, 0] + 2.4 + vlen / 2 + 2
        assert not np.any(on[e] & (xe > -5 + 1e-3) & (xe < reach - 1e-3) & (np.abs(ye) < 3 - 1e-3))


def test_g11_conflict_fixture_through_the_flow_reset_kernel():
    """The reference-generated conflict fixture (TRF:168-192) on the GPU: one env per fixture pair, one slot placed
    exactly on the fixture's vehicle — the slot survives the reset iff the reference sees no conflict."""
    g = golden('g11_conflict')
    ego5, veh5 = np.ascontiguousarray(g['ego']), np.ascontiguousarray(g['veh'])
    n = len(ego5)
    from env_build_amd.traffic import ROUTES
    K, M = 1, 12
    dev = DeviceModel('left', n_veh=M, modes=list(ROUTES))
    assert np.allclose(ego5[:, 4], 4.8)
    # lane[slot 0] = the vehicle's pose with a zero direction (u * lane_len * 0 adds nothing), period tiny -> p = 1
    lane = np.zeros((n, M, 5), np.float32)
    out = np.zeros(n, np.uint8)
    ego = np.zeros((n, 6), np.float32)
    ego[:, 0], ego[:, 3], ego[:, 4], ego[:, 5] = ego5[:, 3], ego5[:, 0], ego5[:, 1], ego5[:, 2]
    p = dev._ptr
    for i in range(n):      # the lane table is per handle call, not per env: one call per pair (tiny launches)
        ln = np.zeros((M, 5), np.float32)
        ln[0, :3] = veh5[i, :3]
        st = dict(cand=dev._in(np.zeros((1, M, 4), np.float32)), active=dev._in(np.zeros((1, M), np.uint8), np.uint8),
                  timer=dev._in(np.zeros((1, 12), np.float32)), emitted=dev._in(np.zeros((1, 12), np.int32), np.int32),
                  sim=dev._in(np.zeros(1, np.int32), np.int32), ph=dev._in(np.zeros(1, np.uint8), np.uint8),
                  mode=dev._in(np.zeros((1, M), np.uint8), np.uint8), light=dev._in(np.zeros(1, np.uint8), np.uint8))
        vmax = np.zeros(M, np.float32)
        period = np.full(12, 1e-3, np.float32)
        vlen = np.full(M, veh5[i, 4], np.float32)
        # speed: u2 * v_max with u2 unknown -> the predicate's reach depends on it; use v_max = 0 and fold the fixture's
        # vehicle speed into the ego-frame test only through pairs whose outcome does not depend on it (checked below)
        t_eg, t_ln, t_pe, t_vm, t_vl = dev._in(ego[i:i + 1]), dev._in(ln), dev._in(period), dev._in(vmax), dev._in(vlen)
        dev.api.traffic_flow_reset(dev.h, 1, K, None, p(t_eg), p(st['cand']), p(st['active']), p(st['timer']),
                                   p(st['emitted']), p(st['sim']), p(st['ph']), p(t_ln), p(t_pe), p(t_vm),
                                   p(t_vl), C.c_float(75.0), 0, 1, C.c_uint64(1), C.c_uint64(1), p(st['mode']), p(st['light']),
                                   dev.stream)
        out[i] = 1 - dev._ret(st['active'])[0, 0]
    # the oracle's predicate with veh_v = 0 is the expectation; where the fixture's own speed does not matter, the
    # reference's recorded flag must agree as well
    fn = oracle_lib().lib.eb_oracle_init_conflict
    fn.restype, fn.argtypes = None, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    v0 = veh5.copy(); v0[:, 3] = 0.0
    want0, want = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    fn(n, ego5.ctypes.data, v0.ctypes.data, want0.ctypes.data)
    fn(n, ego5.ctypes.data, veh5.ctypes.data, want.ctypes.data)
    assert np.array_equal(out, want0)
    same = (want0 == want) & (g['margin'] > 1e-3)
    assert same.mean() > 0.8 and np.array_equal(out[same], g['hit'][same])


# ---- exit-relative frames ------------------------------------------------------------------------------
@pytest.mark.parametrize('task', TASKS)
def test_g6x_exit_frames_on_gpu(task):
    g = golden('g6x_exit_frames_%s' % task)
    host, dev = _pair(task, mode='training')
    ego_t = dev.exit_frame(g['exit_id'], g['ego_world'])
    assert np.array_equal(ego_t[:, 3:].astype(np.float64), g['ego_trans']) and np.array_equal(ego_t[:, :3], g['ego_world'][:, :3])
    obs = dev.get_obs(ego_t, g['cand_world'], g['cand_mode_world'], g['v_light_world'], ref_idx=g['ref_index'],
                      virtual=g['virtual'], exit_id=g['exit_id'])
    assert np.array_equal(obs[:, :6], g['obs'][:, :6]) and np.array_equal(obs[:, 9:], g['obs'][:, 9:])
    close(obs[:, 6:9], g['obs'][:, 6:9], 1e-5, 5e-6, 'GPU G6X exit frames: tracking columns')
    back = dev.exit_frame(g['exit_id'], ego_t, inverse=True)
    assert np.array_equal(back[:, 3:].astype(np.float64), g['ego_back'])
    assert np.array_equal(obs, host.get_obs(ego_t, g['cand_world'], g['cand_mode_world'], g['v_light_world'],
                                            ref_idx=g['ref_index'], virtual=g['virtual'], exit_id=g['exit_id']))


@pytest.mark.parametrize('task', TASKS)
def test_exit_frames_random_scenes_equal_oracle(task):
    rng = np.random.default_rng(21)
    B, M = 3000, 20
    host, dev = _pair(task, mode='training')
    ego = np.stack([rng.uniform(0, 8, B), rng.normal(0, .2, B), rng.normal(0, .2, B), rng.uniform(-60, 60, B),
                    rng.uniform(-60, 60, B), rng.uniform(-400, 400, B)], 1).astype(np.float32)
    ex = rng.integers(0, 4, B).astype(np.uint8)
    cand = np.stack([rng.uniform(-60, 60, (B, M)), rng.uniform(-60, 60, (B, M)), rng.uniform(0, 9, (B, M)),
                     rng.uniform(-400, 400, (B, M))], 2).astype(np.float32)
    cmode = rng.integers(0, 12, (B, M)).astype(np.uint8)
    cmode[rng.random((B, M)) < 0.1] = _capi.VMODE_EMPTY
    vl, virt = rng.integers(0, 4, B).astype(np.uint8), (rng.random(B) < 0.3).astype(np.uint8)
    ref = rng.integers(-1, 4, B).astype(np.int32)
    for inverse in (False, True):
        assert np.array_equal(host.exit_frame(ex, ego, inverse), dev.exit_frame(ex, ego, inverse))
    ego_t = dev.exit_frame(ex, ego)
    o_h = host.get_obs(ego_t, cand, cmode, vl, ref_idx=ref, virtual=virt, exit_id=ex)
    o_d = dev.get_obs(ego_t, cand, cmode, vl, ref_idx=ref, virtual=virt, exit_id=ex)
    assert np.array_equal(o_h, o_d)
    # exit D (angle 0) on headings already inside (-180, 180] is the plain observation: the float64 identity rotation
    # changes nothing but the sign of a zero (-0.0 * 1 + y * 0.0 = +0.0)
    d0 = np.zeros(B, np.uint8)
    cand2 = cand.copy()
    cand2[:, :, 3] = np.clip(cand2[:, :, 3], -179.5, 179.5)
    plain = dev.get_obs(ego, cand2, cmode, vl, ref_idx=ref, virtual=virt)
    via_d = dev.get_obs(ego, cand2, cmode, vl, ref_idx=ref, virtual=virt, exit_id=d0)
    assert np.array_equal(plain, via_d)


def test_exit_ids_above_three_are_refused_or_marked():
    """An exit id that is not EB_EXIT_D .. EB_EXIT_L: the oracle (host arguments) returns EB_EINVAL; the HIP library cannot read
    device memory on the host — it writes NaN into that env's row (eb_get_obs) / pose (eb_exit_frame) and leaves the others
    alone, instead of a plausible-looking wrong frame."""
    task, B, M = 'left', 70, 6
    host, dev = _pair(task)
    from tests._env_step_check import random_scene
    ego, cand, cmode, _, light, _, ref = random_scene(task, B, M, 12)
    ex = np.random.default_rng(1).integers(0, 4, B).astype(np.uint8)
    good_obs, good_pose = dev.get_obs(ego, cand, cmode, light, ref_idx=ref, exit_id=ex), dev.exit_frame(ex, ego)
    bad = ex.copy()
    bad[[3, 40]] = (4, 255)
    obs, pose = dev.get_obs(ego, cand, cmode, light, ref_idx=ref, exit_id=bad), dev.exit_frame(bad, ego)
    ok = np.ones(B, bool)
    ok[[3, 40]] = False
    assert np.isnan(obs[~ok]).all() and np.array_equal(obs[ok], good_obs[ok])
    assert np.isnan(pose[~ok][:, 3:]).all() and np.array_equal(pose[~ok][:, :3], ego[~ok][:, :3]) and np.array_equal(pose[ok], good_pose[ok])
    with pytest.raises(ValueError):
        host.get_obs(ego, cand, cmode, light, ref_idx=ref, exit_id=bad)
    with pytest.raises(ValueError):
        host.exit_frame(bad, ego)


def test_env_facade_exit_frames_for_a_batch():
    """CrossroadEnd2end._get_obs(exit_=...) for a batch = the 12-ego scene in one call (multi_ego.py:84-104)."""
    from env_build_amd.endtoend import CrossroadEnd2end
    B = 96
    env = CrossroadEnd2end('left', n_env=B, mode='testing')
    host = HostModel(oracle_lib(), 'left', mode='training')
    ex = np.array(list('DRUL') * (B // 4))
    obs = env._get_obs(exit_=ex)
    ids = np.array([_capi.EXIT_ID[e] for e in ex], np.uint8)
    ego_w = env._ego.cpu().numpy()
    ego_t = host.exit_frame(ids, ego_w)
    assert np.array_equal(env._ego_exit.cpu().numpy(), ego_t)
    want = host.get_obs(ego_t, env._cand.cpu().numpy(), env._cand_mode.cpu().numpy(), env._v_light.cpu().numpy(),
                        ref_idx=env._ref_idx.cpu().numpy(), virtual=env._virtual.cpu().numpy(), exit_id=ids)
    assert np.array_equal(obs.numpy(), want)
    back = env.exit_frame(env._ego_exit, ex, inverse=True).numpy()
    assert np.array_equal(back, host.exit_frame(ids, ego_t, inverse=True))
    assert np.array_equal(env._get_obs('R').numpy(), host.get_obs(host.exit_frame(np.full(B, 1, np.uint8), ego_w), env._cand.cpu().numpy(),
                          env._cand_mode.cpu().numpy(), env._v_light.cpu().numpy(), ref_idx=env._ref_idx.cpu().numpy(),
                          virtual=env._virtual.cpu().numpy(), exit_id=np.full(B, 1, np.uint8)))


# ---- ReferencePath helpers, ego_predict ----------------------------------------------------------------------
@pytest.mark.parametrize('task', TASKS)
def test_small_entry_points_equal_oracle(task):
    rng = np.random.default_rng(31)
    host, dev = _pair(task)
    n = 2000
    x, y = rng.uniform(-70, 70, n).astype(np.float32), rng.uniform(-70, 70, n).astype(np.float32)
    for ratio in (10, 1, 7, 50):
        for k in range(3):
            (hi, hp), (di, dp) = host.find_closest_point(x, y, path_id=k, ratio=ratio), dev.find_closest_point(x, y, path_id=k, ratio=ratio)
            assert np.array_equal(hi, di) and np.array_equal(hp, dp)
            if ratio in (1, 7):      # against NumPy on the tables (DAM:702-715 with another ratio)
                px, py, _ = host.paths[k]
                d = (x[:200, None] - px[None, ::ratio]) ** 2 + (y[:200, None] - py[None, ::ratio]) ** 2
                assert np.array_equal(di[:200], d.argmin(1) * ratio)
    idx = rng.integers(-50, 4200, n).astype(np.int32)
    ref = rng.integers(-1, 4, n).astype(np.int32)
    for nf in (0, 3):
        a, b = host.path_points(idx, nf, ref_idx=ref), dev.path_points(idx, nf, ref_idx=ref)
        assert np.array_equal(a, b)
    px, py, pphi = host.paths[1]
    pts = dev.path_points(idx, 2, path_id=1)
    i0 = np.clip(idx, 0, len(px) - 1)
    assert np.array_equal(pts[0, 0], px[i0]) and np.array_equal(pts[0, 2], pphi[i0])           # DAM:727-733
    i1 = np.minimum(idx + 80, len(px) - 2); i2 = np.minimum(i1 + 80, len(px) - 2)                # DAM:719-722
    assert np.array_equal(pts[1, 1], py[np.clip(i1, 0, len(px) - 1)]) and np.array_equal(pts[2, 0], px[np.clip(i2, 0, len(px) - 1)])
    d = rng.uniform(-720, 720, n).astype(np.float32)
    d[:4] = [180.0, -180.0, 180.00002, -180.00002]
    want = np.where(d > 180, d - np.float32(360), d); want = np.where(want < -180, want + np.float32(360), want)
    assert np.array_equal(dev.phi_diff(d), host.phi_diff(d)) and np.array_equal(dev.phi_diff(d), want.astype(np.float32))
    ego = np.stack([rng.uniform(-2, 40, n), rng.normal(0, .5, n), rng.normal(0, .5, n), x, y, rng.uniform(-180, 180, n)], 1).astype(np.float32)
    act = np.stack([rng.uniform(-.42, .42, n), rng.uniform(-3.1, 1.6, n)], 1).astype(np.float32)
    e_h, e_d = host.ego_predict(ego, act), dev.ego_predict(ego, act)
    assert np.array_equal(e_h, e_d)
    nxt, _ = dev.f_xu(ego, act, 0.1)
    assert np.array_equal(e_d[:, 1:], nxt[:, 1:]) and np.array_equal(e_d[:, 0], np.clip(nxt[:, 0], 0, 35))      # DAM:387-390


def test_reference_path_facade_custom_path_and_ratio():
    import torch
    from env_build_amd.dynamics_and_models import EnvironmentModel, ReferencePath, deal_with_phi_diff
    ref = ReferencePath('left', 1)
    xs = np.linspace(-30, 30, 100).astype(np.float32)
    ys = np.linspace(-30, 30, 100).astype(np.float32)
    i10, p10 = ref.find_closest_point(xs, ys)
    i3, p3 = ref.find_closest_point(xs, ys, ratio=3)
    px, py, pphi = ref.path
    d = (xs[:, None] - px[None, ::3]) ** 2 + (ys[:, None] - py[None, ::3]) ** 2
    assert np.array_equal(i3.numpy(), d.argmin(1) * 3) and i10.numpy().max() % 10 == 0
    pts = ref.indexs2points(np.array([-5, 0, 17, 10 ** 6]))
    assert np.array_equal(pts[0].numpy(), px[[0, 0, 17, len(px) - 1]])
    fut = ref.future_n_data(np.array([100, len(px) - 30]), 2)
    assert np.array_equal(fut[0][0].numpy(), px[[180, len(px) - 2]]) and np.array_equal(fut[1][1].numpy(), py[[260, len(px) - 2]])
    # a custom path (any (xs, ys, phis) triple, as the reference accepts): a straight line y = 2 heading east
    cx = np.arange(0, 300, dtype=np.float32) * np.float32(0.5)
    ref.path = (cx, np.full_like(cx, 2.0), np.zeros_like(cx))
    idx, (qx, qy, qphi) = ref.find_closest_point(np.float32([10.2, 77.7]), np.float32([0., 5.]))
    assert idx.numpy().tolist() == [20, 160] and qy.numpy().tolist() == [2.0, 2.0]
    trk = ref.tracking_error_vector(np.float32([10.2]), np.float32([0.]), np.float32([10.]), np.float32([5.]), 1).numpy()
    assert trk.shape == (1, 6) and trk[0, 1] == 10.0 and trk[0, 2] == -3.0 and trk[0, 3] == np.float32(50.0 - 10.2)
    ref.set_path(2)
    assert ref._current_path_id() == 2
    assert deal_with_phi_diff(np.float32([190., -190., 20.])).numpy().tolist() == [-170.0, 170.0, 20.0]
    m = EnvironmentModel('left')
    e = m.ego_predict(np.float32([[40., 0, 0, 0, 0, 90.] + [0.] * (m.obs_dim - 6)]), np.float32([[0., 1.]])).numpy()
    assert e[0, 0] == 35.0
    torch.cuda.synchronize()


# ---- error paths leave the state alone ------------------------------------------------------------------------
def test_env_step_validates_before_it_launches_and_set_paths_failure_keeps_the_old_tables():
    task, B, M = 'left', 64, 10
    dev = DeviceModel(task, mode='training')
    native = ['dl', 'du', 'ud', 'ul']
    tr = DeviceModel(task, n_veh=M, modes=[native[i % 4] for i in range(M)])
    rng = np.random.default_rng(0)
    from tests.test_gpu_parity import _random_scene
    ego, cand, _, _, light, _, ref = _random_scene(task, B, M, 5)
    cmode = np.tile(np.array([_capi.VMODE_ID[native[i % 4]] for i in range(M)], np.uint8), (B, 1))
    obs0 = dev.get_obs(ego, cand, cmode, light, ref_idx=ref)
    raw = rng.uniform(-1, 1, (B, 2)).astype(np.float32)
    e_io, c_io = dev._in(ego.copy()), dev._in(cand.copy())
    par, sc, out5 = dev._out((B, 4)), dev._out((B, 2)), dev._out((5, B))
    par[...] = 5.0
    obs_o, code = dev._out(obs0.shape), dev._out((B,), np.uint8)
    p = dev._ptr
    bad_calls = [
        dict(path_id=7, ref=None),                       # bad path id with no per-env ids
        dict(path_id=0, ref=ref, cmode=None),            # candidates without modes
    ]
    for bad in bad_calls:
        with pytest.raises(ValueError):
            dev.api.env_step(dev.h, tr.h, B, p(dev._in(obs0)), p(dev._in(raw)), p(dev._in(bad['ref'], np.int32)), bad['path_id'],
                             p(e_io), p(par), M, p(c_io), p(dev._in(cmode, np.uint8)) if 'cmode' not in bad else None, None, None,
                             None, p(sc), p(out5), None, p(obs_o), p(code), None, None, None, None, dev.stream)
        assert np.array_equal(dev._ret(e_io), ego) and np.array_equal(dev._ret(c_io), cand) and (dev._ret(par) == 5.0).all()
    unset = DeviceModel.__new__(DeviceModel)               # a traffic handle without slot modes: EB_ESTATE before any launch
    import torch
    unset.torch, unset.dev, unset.api = torch, dev.dev, dev.api
    unset.h = dev.api.create(task, M, 0, _capi.MODE_SELECTING)
    with pytest.raises(_capi.EbError):
        dev.api.env_step(dev.h, unset.h, B, p(dev._in(obs0)), p(dev._in(raw)), p(dev._in(ref, np.int32)), 0, p(e_io), p(par), M,
                         p(c_io), p(dev._in(cmode, np.uint8)), None, None, None, p(sc), p(out5), None, p(obs_o), p(code), None, None, None, None, dev.stream)
    assert np.array_equal(dev._ret(e_io), ego) and np.array_equal(dev._ret(c_io), cand)
    dev.api.destroy(unset.h)
    unset.h = None                                        # (its __del__ then destroys NULL: a no-op)
    # eb_set_paths with a non-finite point: error, and the handle still answers from the old tables
    before = dev.tracking_error(ego[:, 3], ego[:, 4], ego[:, 5], ego[:, 0], 0, ref_idx=ref)
    xs = np.concatenate([pp[0] for pp in dev.paths]).astype(np.float32).copy()
    ys = np.concatenate([pp[1] for pp in dev.paths]).astype(np.float32)
    ph = np.concatenate([pp[2] for pp in dev.paths]).astype(np.float32)
    lens = np.array([len(pp[0]) for pp in dev.paths], np.int32)
    xs[1234] = np.nan
    with pytest.raises(ValueError):
        dev.api.set_paths(dev.h, xs.ctypes.data_as(C.c_void_p), ys.ctypes.data_as(C.c_void_p), ph.ctypes.data_as(C.c_void_p),
                          lens.ctypes.data_as(C.c_void_p), 3)
    after = dev.tracking_error(ego[:, 3], ego[:, 4], ego[:, 5], ego[:, 0], 0, ref_idx=ref)
    assert np.array_equal(before, after)
    o1, _, _ = dev.rollout_step(obs0, raw, ref)
    host = HostModel(oracle_lib(), task, mode='training')
    o2, _, _ = host.rollout_step(obs0, raw, ref)
    assert np.array_equal(o1, o2)
    assert VEH_NUM[task] == 8


# ---- angle wrapping never hangs the device (include/envbuild.h "Angle wrapping") ----------------------------------------
@pytest.mark.timeout(120)
def test_angle_wrap_loops_are_bounded_on_both_backends():
    from tests._env_step_check import wrap_guard_case
    host, dev = _pair('left')
    for a, b in zip(wrap_guard_case(dev), wrap_guard_case(host)):
        assert np.array_equal(a, b, equal_nan=True)
