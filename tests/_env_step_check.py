"""CrossroadEnd2end.step as one C-ABI call (eb_env_step) against the six (seven) single calls — the check shared by the
CPU suite (oracle composite vs oracle single calls) and the GPU suite (the one-launch kernel vs the HIP single calls, and
HIP vs oracle)."""
import numpy as np

from env_build_amd import _capi
from env_build_amd.endtoend_env_utils import VEHICLE_MODE_LIST
from env_build_amd.synthetic import make_rollout_inputs

CASES = [('left', 500, 14, None, 0), ('straight', 500, 14, None, 0), ('right', 500, 14, None, 0), ('left', 64, 16, None, 0),
         ('left', 129, 1, None, 0), ('straight', 200, 33, 16, 0), ('right', 70, 64, None, 2), ('left', 100, 64, 64, 0)]


def random_scene(task, B, M, seed):
    rng = np.random.default_rng(seed)
    inp = make_rollout_inputs(task, B, 8, 1, seed=seed)
    ego = inp['ego'].copy()
    ego[:, 1] = rng.normal(0, 0.3, B); ego[:, 2] = rng.normal(0, 0.4, B)
    ego[::9, 3] += rng.uniform(-12, 12, len(ego[::9]))          # some egos off the road
    cand = np.stack([rng.uniform(-60, 60, (B, M)), rng.uniform(-60, 60, (B, M)), rng.uniform(0, 9, (B, M)),
                     rng.uniform(-180, 180, (B, M))], 2).astype(np.float32)
    near = rng.random((B, M)) < 0.01                              # some candidates on top of the ego
    cand[near, 0] = (ego[:, 3][:, None] + rng.uniform(-4, 4, (B, M)))[near]
    cand[near, 1] = (ego[:, 4][:, None] + rng.uniform(-4, 4, (B, M)))[near]
    cmode = rng.integers(0, 12, (B, M)).astype(np.uint8)
    cmode[rng.random((B, M)) < 0.1] = _capi.VMODE_EMPTY
    lw = np.stack([rng.uniform(3.5, 6, (B, M)), rng.uniform(1.6, 2.6, (B, M))], 2).astype(np.float32)
    light = (rng.random(B) < 0.3).astype(np.uint8)
    act = np.stack([rng.uniform(-.42, .42, B), rng.uniform(-3.2, 1.7, B)], 1).astype(np.float32)
    return ego, cand, cmode, lw, light, act, inp['ref_idx']


def composite_case(make, task, B, M, NV, nf, tile=None):
    """make(task, **kw) -> a HostModel / DeviceModel; -> the composite's eight outputs (for cross-library comparison)"""
    native = VEHICLE_MODE_LIST[task]
    modes = [native[i % len(native)] for i in range(M)]
    ego, cand, _, _, light, _, ref = random_scene(task, B, M, 44)
    cmode = np.tile(np.array([_capi.VMODE_ID[m] for m in modes], np.uint8), (B, 1))
    rng = np.random.default_rng(2)
    cmode[rng.random((B, M)) < 0.05] = _capi.VMODE_EMPTY
    raw = rng.uniform(-1.2, 1.2, (B, 2)).astype(np.float32)
    # per-candidate (l, w) as the flow source's vTypes give them (TRF:263-295 reads veh['l'], veh['w']); a third of the
    # candidates sit next to the ego so that the collision outcome depends on them
    lw = np.stack([rng.choice([4.754264, 4.173896, 4.8], (B, M)), rng.choice([1.596668, 1.77515, 2.0, 2.4], (B, M))], 2).astype(np.float32)
    close = rng.random((B, M)) < 0.33
    ang, dist = rng.uniform(-np.pi, np.pi, (B, M)), rng.uniform(1.5, 3.6, (B, M))
    cand = cand.copy()
    cand[:, :, 0] = np.where(close, ego[:, 3:4] + dist * np.cos(ang), cand[:, :, 0])
    cand[:, :, 1] = np.where(close, ego[:, 4:5] + dist * np.sin(ang), cand[:, :, 1])
    gone = rng.random((B, M)) < 0.1                                  # some have left the map: the re-entry rule's business
    cand[:, :, 0] = np.where(gone, rng.choice([-70.0, 66.0, 64.9], (B, M)), cand[:, :, 0]).astype(np.float32)
    virtual = (rng.random(B) < 0.3).astype(np.uint8)
    v_light = rng.integers(0, 4, B).astype(np.uint8)
    entry = np.stack([rng.uniform(-60, 60, M), rng.uniform(-60, 60, M), rng.choice([0., 90., 180., -90.], M),
                      rng.choice([-1., 0., 1.], M), rng.choice([-1., 0., 1.], M)], 1).astype(np.float32)
    rule = dict(entry=entry, limit=65.0, span=60.0, v_max=8.0, seed=0x1234567, counter=9)
    kw = dict(mode='training', n_future=nf)
    if NV is not None:
        kw.update(n_veh=NV)
    m, tr = make(task, **kw), make(task, n_veh=M, modes=modes)
    if tile is not None:
        m.set_tile(tile)          # the one-launch step: 0 / 1 / 2 = 64- / 32- / 16-env tiles (eb_debug_set_tile)
    obs0 = m.get_obs(ego, cand, cmode, light, ref_idx=ref)
    # the six calls
    act = m.action_transform(raw)
    o5, d16 = m.compute_rewards(obs0, act)
    ego1, par1 = m.env_ego_step(ego, act)
    cand1 = tr.veh_predict(cand.reshape(B, -1)).reshape(B, M, 4)
    obs1 = m.get_obs(ego1, cand1, cmode, v_light, ref_idx=ref, virtual=virtual)
    done1 = m.judge_done(ego1, par1, obs1, cand1, cmode, lw, v_light)
    done_default = m.judge_done(ego1, par1, obs1, cand1, cmode, None, v_light)
    if B >= 500:
        assert (done1 != done_default).any()   # the (l, w) pairs matter in this scene (4.8 x 2.0 is not assumed)
    # the composite (state updated in place)
    got = m.env_step(tr, obs0, raw, ego, cand, cmode, ref_idx=ref, cand_lw=lw, v_light=v_light, virtual=virtual)
    want = [act, o5, d16, ego1, par1, cand1, obs1, done1]
    for k, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(np.asarray(g).reshape(np.asarray(w).shape), w), k
    # + the seventh call, and the nullable outputs left out
    cand2, flags = tr.traffic_respawn(cand1, entry, 65.0, 60.0, 8.0, 0x1234567, 9)
    assert flags.any() and not flags.all() and not np.array_equal(cand2, cand1)
    got7 = m.env_step(tr, obs0, raw, ego, cand, cmode, ref_idx=ref, cand_lw=lw, v_light=v_light, virtual=virtual,
                      respawn=rule, want_scaled=False, want_dict=False)
    assert got7[0] is None and got7[2] is None
    for k, (g, w) in enumerate(zip(got7, [None, o5, None, ego1, par1, cand2, obs1, done1])):
        if w is not None:
            assert np.array_equal(np.asarray(g).reshape(np.asarray(w).shape), w), k
    # the scaled actions may overwrite the raw ones (the kernel stores them after every wave has read the raw pair)
    inpl = m.env_step(tr, obs0, raw, ego, cand, cmode, ref_idx=ref, cand_lw=lw, v_light=v_light, virtual=virtual, scale_in_place=True)
    for k, (g, w) in enumerate(zip(inpl, want)):
        assert np.array_equal(np.asarray(g).reshape(np.asarray(w).shape), w), ('in place', k)
    return got


def auto_reset_case(make, task, B, M, NV=None, nf=0, tile=None, close_p=0.02, seed=5, v_light_none=False, strict=True):
    """eb_env_step(auto_reset) == eb_env_step, then the terminal rows -> final_obs, then eb_env_reset_pool(mask = done != 0) in place —
    every output and every piece of state, bit for bit; done_code keeps the step's codes."""
    from env_build_amd.endtoend import _lane_entry
    native = VEHICLE_MODE_LIST[task]
    modes = [native[i % len(native)] for i in range(M)]
    entry = np.array([list(_lane_entry(m)[:3]) + list(_lane_entry(m)[3]) for m in modes], np.float32)
    ego, cand, _, _, _, _, ref = random_scene(task, B, M, seed)
    cmode = np.tile(np.array([_capi.VMODE_ID[m] for m in modes], np.uint8), (B, 1))
    rng = np.random.default_rng(seed + 1)
    cmode[rng.random((B, M)) < 0.05] = _capi.VMODE_EMPTY
    raw = rng.uniform(-1.2, 1.2, (B, 2)).astype(np.float32)
    close = rng.random((B, M)) < close_p                                # a few candidates on the ego: collisions
    ang, dist = rng.uniform(-np.pi, np.pi, (B, M)), rng.uniform(1.5, 3.6, (B, M))
    cand = cand.copy()
    cand[:, :, 0] = np.where(close, ego[:, 3:4] + dist * np.cos(ang), cand[:, :, 0])
    cand[:, :, 1] = np.where(close, ego[:, 4:5] + dist * np.sin(ang), cand[:, :, 1])
    gone = rng.random((B, M)) < 0.1
    cand[:, :, 0] = np.where(gone, rng.choice([-70.0, 66.0, 64.9], (B, M)), cand[:, :, 0]).astype(np.float32)
    virtual = (rng.random(B) < 0.4).astype(np.uint8)
    v_light = None if v_light_none else rng.integers(0, 3, B).astype(np.uint8)
    rule = dict(entry=entry, limit=65.0, span=5.0, v_max=8.0, seed=0x1234567, counter=9)
    pool = dict(entry=entry, span=60.0, v_max=8.0, seed=4242, counter=17, edge_span=5.0)
    kw = dict(mode='training', n_future=nf)
    if NV is not None:
        kw.update(n_veh=NV)
    m, tr = make(task, **kw), make(task, n_veh=M, modes=modes)
    if tile is not None:
        m.set_tile(tile)
    obs0 = m.get_obs(ego, cand, cmode, v_light, ref_idx=ref, virtual=virtual)
    # the two calls
    sc, o5, d16, ego1, par1, cand1, obs1, done1 = m.env_step(tr, obs0, raw, ego, cand, cmode, ref_idx=ref, v_light=v_light, virtual=virtual,
                                                             respawn=rule)
    fin = done1 != 0
    assert not strict or 0.02 < fin.mean() < 0.9, fin.mean()       # (strict: the case must exercise both kinds of row)
    vl_in = np.zeros(B, np.uint8) if v_light is None else v_light
    e2, p2, r2, vf2, vl2, _, c2, o2 = m.env_reset_pool(tr, 99, 5, 1, ego1, par1, ref, virtual, vl_in, cand1, cmode, obs1, pool,
                                                       mask=fin.astype(np.uint8))
    final = np.where(fin[:, None], obs1, np.float32(np.nan))
    # the one call
    got = m.env_step(tr, obs0, raw, ego, cand, cmode, ref_idx=ref, v_light=v_light, virtual=virtual, respawn=rule,
                     auto_reset=dict(seed=99, counter=5, training=1, pool=pool))
    want = [sc, o5, d16, e2, p2, c2, o2, done1, r2, vf2, None if v_light is None else vl2, final]
    names = ['scaled', 'out5', 'dict16', 'ego', 'params', 'cand', 'obs', 'done', 'ref_idx', 'virtual', 'v_light', 'final_obs']
    for k, (g, w) in enumerate(zip(got, want)):
        if w is None:
            assert g is None, names[k]
            continue
        assert np.array_equal(np.asarray(g).reshape(np.asarray(w).shape), w, equal_nan=True), names[k]
    # the rows that did not finish are the plain step's; those that did start from a drawn state
    assert np.array_equal(got[6][~fin], obs1[~fin]) and np.array_equal(got[3][~fin], ego1[~fin])
    assert (got[3][fin][:, 1:3] == 0).all() and (not fin.any() or not np.array_equal(got[6][fin], obs1[fin]))
    # final_obs left out; auto_reset without the step's own re-entry rule
    g2 = m.env_step(tr, obs0, raw, ego, cand, cmode, ref_idx=ref, v_light=v_light, virtual=virtual, respawn=rule,
                    auto_reset=dict(seed=99, counter=5, training=1, pool=pool, final_obs=False))
    assert g2[11] is None
    for k in range(11):
        if want[k] is not None:
            assert np.array_equal(g2[k], got[k]), names[k]
    return got


def auto_reset_bad_args_case(make, task='left', B=40, M=8):
    """auto_reset's pointers must be the call's own arrays, final_obs an array of its own: EB_EINVAL, state untouched."""
    import pytest
    from env_build_amd.endtoend import _lane_entry
    native = VEHICLE_MODE_LIST[task]
    modes = [native[i % len(native)] for i in range(M)]
    entry = np.array([list(_lane_entry(m)[:3]) + list(_lane_entry(m)[3]) for m in modes], np.float32)
    ego, cand, _, _, light, _, ref = random_scene(task, B, M, 3)
    cmode = np.tile(np.array([_capi.VMODE_ID[m] for m in modes], np.uint8), (B, 1))
    m, tr = make(task, mode='training'), make(task, n_veh=M, modes=modes)
    obs0 = m.get_obs(ego, cand, cmode, light, ref_idx=ref)
    raw = np.zeros((B, 2), np.float32)
    pool = dict(entry=entry, span=60.0, v_max=8.0, seed=1, counter=1, edge_span=5.0)
    virtual = np.zeros(B, np.uint8)
    for wrong in ('ref_idx', 'virtual', 'v_light'):
        with pytest.raises(ValueError):
            m.env_step(tr, obs0, raw, ego, cand, cmode, ref_idx=ref, v_light=light, virtual=virtual,
                       auto_reset=dict(seed=1, counter=1, training=1, pool=pool, wrong=(wrong,)))
    with pytest.raises(ValueError):      # no per-env path ids: a reset cannot record its drawn path
        m.env_step(tr, obs0, raw, ego, cand, cmode, ref_idx=None, v_light=light, virtual=virtual,
                   auto_reset=dict(seed=1, counter=1, training=1, pool=pool))


def flow_rule_case(make, task, B=300, K=5, steps=40, tile=None, light_cycle=1, seed=3, strict=True, hostile=False):
    """eb_env_step(flow) == eb_env_step, then eb_traffic_flow_step on what it left — every output and every piece of the flow
    source's state, bit for bit, over a closed loop that starts from an empty junction (emissions, exits, accelerations, the
    light programme all occur on the way); -> the trace of the fused path (for cross-library comparison)."""
    from env_build_amd.traffic import ACCEL, EXIT_RANGE, FLOWS, LANE_START, ROUTES, VTYPES, approach_lane
    M = 12 * K
    slot_modes = [r for r in ROUTES for _ in range(K)]
    lane = np.array([list(approach_lane(m)[0]) + list(approach_lane(m)[1]) for m in slot_modes], np.float32)
    period = (np.array([3600.0 / FLOWS[r][0] for r in ROUTES], np.float32) / 8).astype(np.float32)       # (dense traffic: things happen within 40 steps)
    vmax = np.array([VTYPES[FLOWS[m][1]][2] for m in slot_modes], np.float32)
    lw = np.tile(np.array([[VTYPES[FLOWS[m][1]][0], VTYPES[FLOWS[m][1]][1]] for m in slot_modes], np.float32), (B, 1, 1))
    rng = np.random.default_rng(seed)
    inp = make_rollout_inputs(task, B, 8, 1, seed=seed)
    ego, ref = inp['ego'].copy(), inp['ref_idx']
    m, tr = make(task, mode='training'), make(task, n_veh=M, modes=slot_modes)
    if tile is not None:
        m.set_tile(tile)
    # a junction in mid-traffic: some slots occupied along their lanes (a few of them far out and heading away: they leave)
    active = (rng.random((B, M)) < 0.4).astype(np.uint8)
    along = rng.uniform(0, 95, (B, M)).astype(np.float32)
    along = np.where(rng.random((B, M)) < 0.15, rng.uniform(150, 175, (B, M)), along).astype(np.float32)   # through the junction and 50-75 m beyond: leaving
    cand = np.stack([lane[None, :, 0] + along * lane[None, :, 3], lane[None, :, 1] + along * lane[None, :, 4],
                     rng.uniform(0, 9, (B, M)).astype(np.float32), np.broadcast_to(lane[None, :, 2], (B, M))], 2).astype(np.float32)
    if hostile:
        # records no traffic source would produce — what the exit rule's "far out AND heading away" (the sign of x cos + y sin at the
        # record's NEW heading) must still answer exactly: far-out vehicles at any heading, among them headings at a right angle to
        # the position vector (the sum within rounding of zero, either sign), headings of many turns, far-out records on turning
        # slots, huge speeds (from the junction box to beyond the exit range in one step), non-finite fields
        active = (rng.random((B, M)) < 0.8).astype(np.uint8)
        r = rng.uniform(40, 120, (B, M)); th = rng.uniform(-np.pi, np.pi, (B, M))
        x, y = r * np.cos(th), r * np.sin(th)
        kind = rng.integers(0, 8, (B, M))
        phi = np.degrees(th) + np.where(kind == 0, 90.0, np.where(kind == 1, -90.0, rng.uniform(-180, 180, (B, M))))   # kinds 0, 1: tangential
        phi = np.where(kind == 2, np.degrees(th), phi)                                   # radially outward
        phi = np.where(kind == 3, np.degrees(th) + 180.0, phi)                           # radially inward
        phi = np.where(kind == 4, phi + 360.0 * rng.integers(-40, 40, (B, M)), phi)      # many turns
        v = rng.uniform(0, 9, (B, M))
        fast = kind == 5
        x = np.where(fast, rng.uniform(-24, 24, (B, M)), x); y = np.where(fast, rng.uniform(-24, 24, (B, M)), y)
        v = np.where(fast, rng.uniform(500, 2000, (B, M)), v)                            # in the box now, far out after the step
        cand = np.stack([x, y, v, phi], 2).astype(np.float32)
        cand[::7, 3, 0] = np.nan; cand[3::11, 5, 3] = np.inf; cand[5::13, 2, 2] = np.inf; cand[1::17, 4, 1] = -np.inf
        cand[2::5, 1] = (70.0, 0.0, 3.0, 90.0); cand[4::5, 1] = (0.0, -80.0, 3.0, 180.0)  # exactly at a right angle, on the axes
    mode = np.where(active != 0, np.array([_capi.VMODE_ID[x] for x in slot_modes], np.uint8)[None, :], _capi.VMODE_EMPTY).astype(np.uint8)
    timer = (rng.random((B, 12)) * period).astype(np.float32)
    emitted, sim_step = np.zeros((B, 12), np.int32), rng.integers(0, 600, B).astype(np.int32)
    light = rng.integers(0, 4, B).astype(np.uint8)
    virtual = (rng.random(B) < 0.3).astype(np.uint8)
    obs = m.get_obs(ego, cand, mode, light, ref_idx=ref, virtual=virtual)
    const = dict(per_route=K, lane=lane, period=period, v_max=vmax, dt=0.1, exit_range=EXIT_RANGE, accel=ACCEL, lane_len=LANE_START - 25.0,
                 light_cycle=light_cycle, seed=99)
    trace, events = [], dict(emit=0, exit=0)
    for t in range(steps):
        raw = rng.uniform(-0.4, 0.4, (B, 2)).astype(np.float32)
        flow = dict(const, active=active, timer=timer, emitted=emitted, sim_step=sim_step, counter=t + 1)
        # the two calls
        a = m.env_step(tr, obs, raw, ego, cand, mode, ref_idx=ref, cand_lw=lw, v_light=light, virtual=virtual)
        f = tr.traffic_flow_step(K, a[5], active, timer, emitted, sim_step, lane, period, vmax, 0.1, EXIT_RANGE, ACCEL, LANE_START - 25.0,
                                 light_cycle, 99, t + 1, light)
        # the one call
        g = m.env_step(tr, obs, raw, ego, cand, mode, ref_idx=ref, cand_lw=lw, v_light=light, virtual=virtual, flow=flow)
        want = [a[0], a[1], a[2], a[3], a[4], f[0], a[6], a[7], f[1], f[2], f[3], f[4], f[5], f[6]]
        names = ['scaled', 'out5', 'dict16', 'ego', 'params', 'cand', 'obs', 'done', 'active', 'timer', 'emitted', 'sim_step', 'cand_mode', 'v_light']
        for k, (x, y) in enumerate(zip(g, want)):
            assert np.array_equal(np.asarray(x).reshape(np.asarray(y).shape), y, equal_nan=hostile), (t, names[k])
        events['emit'] += int((f[3] != emitted).sum())
        events['exit'] += int(((active != 0) & (f[1] == 0)).sum())
        ego, cand, obs = g[3], g[5], g[6]
        active, timer, emitted, sim_step, mode, light = g[8], g[9], g[10], g[11], g[12], g[13]
        trace.append([np.asarray(x) for x in g])
    assert not strict or (events['emit'] > B and events['exit'] > 0), events
    return trace


def flow_rule_bad_args_case(make, task='left', B=20, K=2):
    """the flow rule excludes the pool's rules; its cand_mode / v_light must be the call's own arrays: EB_EINVAL, nothing written"""
    import pytest
    from env_build_amd.traffic import ROUTES
    M = 12 * K
    slot_modes = [r for r in ROUTES for _ in range(K)]
    m, tr = make(task, mode='training'), make(task, n_veh=M, modes=slot_modes)
    ego, cand, _, _, light, _, ref = random_scene(task, B, M, 3)
    mode = np.tile(np.array([_capi.VMODE_ID[x] for x in slot_modes], np.uint8), (B, 1))
    obs = m.get_obs(ego, cand, mode, light, ref_idx=ref)
    raw = np.zeros((B, 2), np.float32)
    z = lambda *s: np.zeros(s, np.float32)
    flow = dict(per_route=K, active=np.ones((B, M), np.uint8), timer=z(B, 12), emitted=np.zeros((B, 12), np.int32), sim_step=np.zeros(B, np.int32),
                lane=z(M, 5), period=np.ones(12, np.float32), v_max=np.ones(M, np.float32), dt=0.1, exit_range=40.0, accel=1.0, lane_len=60.0,
                light_cycle=1, seed=1, counter=1)
    for bad in (dict(flow, wrong=('mode',)), dict(flow, wrong=('v_light',)), dict(flow, per_route=K + 1)):
        with pytest.raises(ValueError):
            m.env_step(tr, obs, raw, ego, cand, mode, ref_idx=ref, v_light=light, flow=bad)
    rule = dict(entry=z(M, 5), limit=65.0, span=5.0, v_max=8.0, seed=1, counter=1)
    with pytest.raises(ValueError):
        m.env_step(tr, obs, raw, ego, cand, mode, ref_idx=ref, v_light=light, flow=flow, respawn=rule)


def masked_obs_case(make, task, B=150, M=10, seed=8):
    """eb_get_obs with a row mask: the masked rows equal the unmasked call's, the others keep what the buffer held."""
    native = VEHICLE_MODE_LIST[task]
    ego, cand, cmode, _, light, _, ref = random_scene(task, B, M, seed)
    m = make(task, mode='training')
    full = m.get_obs(ego, cand, cmode, light, ref_idx=ref)
    rng = np.random.default_rng(seed)
    for mask in (rng.random(B) < 0.2, np.zeros(B, bool), np.ones(B, bool), np.arange(B) == B - 1):
        init = rng.normal(size=full.shape).astype(np.float32)
        got = m.get_obs(ego, cand, cmode, light, ref_idx=ref, row_mask=mask.astype(np.uint8), obs_init=init)
        assert np.array_equal(got[mask], full[mask]) and np.array_equal(got[~mask], init[~mask])
    return full


def respawn_conflict_case(make, B=300, M=16, seed=4):
    """eb_traffic_respawn with the ego given (init_traffic's conflict rule for the pool, TRF:168-192): a forced re-entry
    puts nobody inside the ego's box; the candidates that would have been are at their lane's edge; everybody else is
    where the rule without the ego puts them."""
    from env_build_amd.endtoend import _lane_entry
    task = 'left'
    native = VEHICLE_MODE_LIST[task]
    modes = [native[i % len(native)] for i in range(M)]
    entry = np.array([list(_lane_entry(m)[:3]) + list(_lane_entry(m)[3]) for m in modes], np.float32)
    ego, cand, _, _, _, _, _ = random_scene(task, B, M, seed)
    ego[:, 1:3] = 0
    tr = make(task, n_veh=M, modes=modes)
    plain, f0 = tr.traffic_respawn(cand, entry, -1.0, 60.0, 8.0, 99, 5)
    safe, f1 = tr.traffic_respawn(cand, entry, -1.0, 60.0, 8.0, 99, 5, ego=ego, edge_span=5.0)
    assert f0.all() and f1.all()
    moved = np.any(plain != safe, axis=2)
    assert 0.005 < moved.mean() < 0.3                                    # some candidates would have started on the ego
    assert np.array_equal(plain[~moved], safe[~moved])
    along = np.abs((safe[..., :2] - entry[None, :, :2]) @ np.array([1.0, 1.0], np.float32))   # lanes are axis-parallel
    assert (along[moved] < 5.0 + 1e-3).all()
    # nobody within 3 m laterally and [-5, v + 6.8] m longitudinally of an ego heading along its lane... checked loosely:
    d = np.hypot(safe[..., 0] - ego[:, None, 3], safe[..., 1] - ego[:, None, 4])
    assert d.min() > 2.0
    return safe


def reset_pool_case(make, task, B=260, M=12, seed=21, tile=None, NV=None):
    """eb_env_reset_pool == eb_env_reset, eb_traffic_respawn(forced, clear of the ego), v_light clear, eb_get_obs(row mask, OLD
    flags), flag swap — for the masked envs only; the other envs' rows are untouched."""
    from env_build_amd.endtoend import _lane_entry
    native = VEHICLE_MODE_LIST[task]
    modes = [native[i % len(native)] for i in range(M)]
    entry = np.array([list(_lane_entry(m)[:3]) + list(_lane_entry(m)[3]) for m in modes], np.float32)
    ego, cand, _, _, _, _, ref = random_scene(task, B, M, seed)
    cmode = np.tile(np.array([_capi.VMODE_ID[m] for m in modes], np.uint8), (B, 1))
    rng = np.random.default_rng(seed)
    params = rng.normal(size=(B, 4)).astype(np.float32)
    virtual = (rng.random(B) < 0.5).astype(np.uint8)
    v_light = rng.integers(0, 4, B).astype(np.uint8)
    nv = len(native) if NV is None else NV
    obs = rng.normal(size=(B, 9 + 4 * nv)).astype(np.float32)
    m = make(task, mode='training') if NV is None else make(task, n_veh=NV, mode='training')
    tr = make(task, n_veh=M, modes=modes)
    if tile is not None:
        m.set_tile(tile)          # the one-launch reset: 0 / 1 / 2 = 64- / 32- / 16-env tiles (eb_debug_set_tile)
    pool = dict(entry=entry, span=60.0, v_max=8.0, seed=4242, counter=17, edge_span=5.0)
    outs = []
    for mask in (rng.random(B) < 0.3, None):
        mk = None if mask is None else mask.astype(np.uint8)
        sel = np.ones(B, bool) if mask is None else mask
        got = m.env_reset_pool(tr, 99, 5, 1, ego, params, ref, virtual, v_light, cand, cmode, obs, pool, mask=mk)
        # the single calls
        e1, p1, r1, vnext, dc = m.env_reset(B, 99, 5, 1, ego, params, ref, mask=mk)
        c1, _ = tr.traffic_respawn(cand, entry, -1.0, 60.0, 8.0, 4242, 17, mask=mk, ego=e1, edge_span=5.0)
        vl1 = np.where(sel, 0, v_light).astype(np.uint8)
        o1 = m.get_obs(e1, c1, cmode, vl1, ref_idx=r1, virtual=virtual, row_mask=mk, obs_init=obs)
        vf1 = np.where(sel, vnext, virtual).astype(np.uint8)
        want = [e1, p1, r1, vf1, vl1, np.where(sel, 0, 7).astype(np.uint8), c1, o1]
        for k, (g, w) in enumerate(zip(got, want)):
            assert np.array_equal(g, w), k
        assert np.array_equal(got[0][~sel], ego[~sel]) and np.array_equal(got[7][~sel], obs[~sel]) and np.array_equal(got[6][~sel], cand[~sel])
        outs.append(got)
        # obs_src / done_src: the rows outside the mask come from the previous arrays, those inside are the same as above
        prev_obs = rng.normal(size=obs.shape).astype(np.float32)
        prev_done = rng.integers(0, 7, B).astype(np.uint8)
        g2 = m.env_reset_pool(tr, 99, 5, 1, ego, params, ref, virtual, v_light, cand, cmode, obs, pool, mask=mk, obs_src=prev_obs, done_src=prev_done)
        for k in (0, 1, 2, 3, 4, 6):
            assert np.array_equal(g2[k], got[k]), k
        assert np.array_equal(g2[7][sel], got[7][sel]) and np.array_equal(g2[7][~sel], prev_obs[~sel])
        assert (g2[5][sel] == 0).all() and np.array_equal(g2[5][~sel], prev_done[~sel])
        outs.append(g2)
    return outs


def wrap_guard_case(m):
    """Headings the reference's `while` wraps (UTL:134-139, 232-237) would never finish on — +-inf, +-1e30, 4e6 degrees — come back
    unwrapped (EB_WRAP_MAX_DEG) instead of hanging; ordinary and just-inside values wrap as always."""
    phis = np.array([0.0, 179.0, 181.0, -540.0, 3.59e6, -3.59e6, 3.6e6, 4.0e6, -4.0e6, 1e30, -1e30, np.inf, -np.inf, np.nan], np.float32)
    n = len(phis)
    ego = np.zeros((n, 6), np.float32)
    ego[:, 0], ego[:, 5] = 5.0, phis
    out = [m.exit_frame(np.full(n, k, np.uint8), ego, inverse=inv)[:, 5] for k in range(4) for inv in (False, True)]
    nxt, _ = m.env_ego_step(ego, np.zeros((n, 2), np.float32))
    out.append(nxt[:, 5])
    big = np.abs(phis) > 3.6e6
    for o in out[:2]:                                   # exit D: the heading itself
        assert np.array_equal(o[big], phis[big], equal_nan=True) and (np.abs(o[~big & np.isfinite(phis)]) <= 180.0).all()
    assert (np.abs(out[-1][:6]) <= 180.0).all()         # the ego step wraps what it can
    return out


def time_limit_case(make, task, B, M, tile=None, max_steps=200, seed=8):
    """ABI 5, eb_time_limit — gym's TimeLimit around the registered env (README.md:55-59): the step counts rise by one; an env no
    reference predicate has finished takes EB_DONE_TIME_LIMIT when its count reaches the limit; a finished env's count restarts;
    nothing else about the step changes.  With auto_reset the truncated envs are reset like the others; the resets clear the
    counts of the rows they touch.  -> the outputs of the limited step + auto reset (for cross-library comparison)."""
    from env_build_amd.endtoend import _lane_entry
    native = VEHICLE_MODE_LIST[task]
    modes = [native[i % len(native)] for i in range(M)]
    entry = np.array([list(_lane_entry(m)[:3]) + list(_lane_entry(m)[3]) for m in modes], np.float32)
    ego, cand, _, _, light, _, ref = random_scene(task, B, M, seed)
    cmode = np.tile(np.array([_capi.VMODE_ID[m] for m in modes], np.uint8), (B, 1))
    rng = np.random.default_rng(seed + 1)
    raw = rng.uniform(-1.2, 1.2, (B, 2)).astype(np.float32)
    virtual = (rng.random(B) < 0.3).astype(np.uint8)
    steps = rng.integers(0, max_steps + 5, B).astype(np.int32)
    steps[::3] = max_steps - 1                                       # a third of the envs are about to be truncated
    steps[1::7] = max_steps - 2                                      # ... and some one step short of it
    m, tr = make(task, mode='training'), make(task, n_veh=M, modes=modes)
    if tile is not None:
        m.set_tile(tile)
    obs0 = m.get_obs(ego, cand, cmode, light, ref_idx=ref)
    plain = m.env_step(tr, obs0, raw, ego, cand, cmode, ref_idx=ref, v_light=light, virtual=virtual)
    got = m.env_step(tr, obs0, raw, ego, cand, cmode, ref_idx=ref, v_light=light, virtual=virtual, time_limit=(steps, max_steps))
    done0 = plain[7]
    want_code = np.where((done0 == 0) & (steps + 1 >= max_steps), np.uint8(7), done0)
    want_steps = np.where(want_code != 0, 0, steps + 1).astype(np.int32)
    assert (want_code == 7).any() and (done0 != 0).any() and (want_code == 0).any()
    for k in range(7):                                               # scaled, out5, dict16, ego, params, cand, obs: the plain step's
        assert np.array_equal(got[k], plain[k]), k
    assert np.array_equal(got[7], want_code) and np.array_equal(got[-1], want_steps)
    # with the reset of the finished envs in the same call: the truncated envs are finished envs
    pool = dict(entry=entry, span=60.0, v_max=8.0, seed=4242, counter=17, edge_span=5.0)
    auto = m.env_step(tr, obs0, raw, ego, cand, cmode, ref_idx=ref, v_light=light, virtual=virtual,
                      auto_reset=dict(seed=99, counter=5, training=1, pool=pool), time_limit=(steps, max_steps))
    fin = want_code != 0
    e2, p2, r2, vf2, vl2, _, c2, o2, s2 = m.env_reset_pool(tr, 99, 5, 1, plain[3], plain[4], ref, virtual, light, plain[5], cmode, plain[6],
                                                           pool, mask=fin.astype(np.uint8), episode_step=want_steps + 3 * (~fin))
    final = np.where(fin[:, None], plain[6], np.float32(np.nan))
    want = [plain[0], plain[1], plain[2], e2, p2, c2, o2, want_code, r2, vf2, vl2, final, want_steps]
    for k, (g, w) in enumerate(zip(auto, want)):
        assert np.array_equal(np.asarray(g).reshape(np.asarray(w).shape), w, equal_nan=True), k
    assert np.array_equal(s2, np.where(fin, 0, want_steps + 3))    # eb_env_reset_pool clears the counts of the rows it resets, only those
    # eb_env_reset likewise
    rs = m.env_reset(B, 5, 6, 1, ego, plain[4], ref, mask=fin.astype(np.uint8), episode_step=steps + 1)
    assert np.array_equal(rs[5], np.where(fin, 0, steps + 1))
    return auto


def parked_ego_case(make, task='left', B=3, max_steps=200):
    """An ego parked on its approach lane (v_x = 0, nothing near it, no light) satisfies no reference predicate — without the step
    limit its episode never ends.  201 closed-loop steps: 'not done yet' for 199 steps, EB_DONE_TIME_LIMIT at step 200, count back
    at 0, step 201 is step 1 of the next count."""
    m, tr = make(task, mode='training'), make(task, n_veh=2, modes=[VEHICLE_MODE_LIST[task][0]] * 2)
    ref = np.zeros(B, np.int32)
    x, y, phi = {'left': (1.875, -40.0, 90.0), 'straight': (5.625, -40.0, 90.0), 'right': (9.375, -40.0, 90.0)}[task]
    ego = np.tile(np.array([0.0, 0.0, 0.0, x, y, phi], np.float32), (B, 1))
    cand = np.zeros((B, 2, 4), np.float32)
    cmode = np.full((B, 2), _capi.VMODE_EMPTY, np.uint8)
    light = np.zeros(B, np.uint8)
    raw = np.tile(np.array([0.0, -1.0], np.float32), (B, 1))        # full brake: v_x stays at its floor (E2E:281)
    steps = np.array([0, 150, 199][:B], np.int32)
    obs = m.get_obs(ego, cand, cmode, light, ref_idx=ref)
    codes = []
    for t in range(max_steps + 1):
        out = m.env_step(tr, obs, raw, ego, cand, cmode, ref_idx=ref, v_light=light, time_limit=(steps, max_steps), want_dict=False)
        ego, obs, code, steps = out[3], out[6], out[7], out[-1]
        codes.append(code.copy())
    codes = np.array(codes)                                           # [201, B]
    assert (ego[:, 0] == 0).all() and np.array_equal(ego[:, 3:], np.tile(np.array([x, y, phi], np.float32), (B, 1)))
    for b, s0 in enumerate([0, 150, 199][:B]):
        hits = np.flatnonzero(codes[:, b])
        first = max_steps - 1 - s0                                    # the step whose count reaches the limit
        want_hits = [first] + ([first + max_steps] if first + max_steps <= max_steps else [])
        assert hits.tolist() == want_hits and (codes[hits, b] == 7).all(), (b, hits)
    return codes


def flow_auto_reset_case(make, task, B=260, K=5, steps=12, tile=None, seed=13, strict=True):
    """ABI 5 — eb_env_step(flow + auto_reset): the step over the flow source that also resets the envs it finished == eb_env_step(flow),
    then the terminal rows -> final_obs, eb_env_reset(mask), eb_traffic_flow_reset(mask, new ego), eb_get_obs(row_mask, OLD flags, the
    light the reset set), flag swap — every output and every piece of state, bit for bit, over a closed loop in which egos do finish;
    -> the trace of the one-call path (for cross-library comparison)."""
    from env_build_amd.traffic import ACCEL, EXIT_RANGE, FLOWS, LANE_START, ROUTES, VTYPES, approach_lane
    M = 12 * K
    slot_modes = [r for r in ROUTES for _ in range(K)]
    lane = np.array([list(approach_lane(m)[0]) + list(approach_lane(m)[1]) for m in slot_modes], np.float32)
    period = (np.array([3600.0 / FLOWS[r][0] for r in ROUTES], np.float32) / 8).astype(np.float32)
    vmax = np.array([VTYPES[FLOWS[m][1]][2] for m in slot_modes], np.float32)
    clen = np.array([VTYPES[FLOWS[m][1]][0] for m in slot_modes], np.float32)
    lw = np.tile(np.array([[VTYPES[FLOWS[m][1]][0], VTYPES[FLOWS[m][1]][1]] for m in slot_modes], np.float32), (B, 1, 1))
    rng = np.random.default_rng(seed)
    inp = make_rollout_inputs(task, B, 8, 1, seed=seed)
    ego, ref = inp['ego'].copy(), inp['ref_idx'].copy()
    ego[::6, 3] += rng.uniform(7, 12, len(ego[::6])) * rng.choice([-1, 1], len(ego[::6]))      # a sixth of the egos off the road: they finish at once
    m, tr = make(task, mode='training'), make(task, n_veh=M, modes=slot_modes)
    if tile is not None:
        m.set_tile(tile)
    active = (rng.random((B, M)) < 0.4).astype(np.uint8)
    along = rng.uniform(0, 95, (B, M)).astype(np.float32)
    cand = np.stack([lane[None, :, 0] + along * lane[None, :, 3], lane[None, :, 1] + along * lane[None, :, 4],
                     rng.uniform(0, 9, (B, M)).astype(np.float32), np.broadcast_to(lane[None, :, 2], (B, M))], 2).astype(np.float32)
    mode = np.where(active != 0, np.array([_capi.VMODE_ID[x] for x in slot_modes], np.uint8)[None, :], _capi.VMODE_EMPTY).astype(np.uint8)
    timer = (rng.random((B, 12)) * period).astype(np.float32)
    emitted, sim_step = np.zeros((B, 12), np.int32), rng.integers(0, 600, B).astype(np.int32)
    light = rng.integers(0, 4, B).astype(np.uint8)
    virtual = (rng.random(B) < 0.3).astype(np.uint8)
    phase0 = np.full(B, 9, np.uint8)
    obs = m.get_obs(ego, cand, mode, light, ref_idx=ref, virtual=virtual)
    rp = 1 if task == 'right' else 0
    const = dict(per_route=K, lane=lane, period=period, v_max=vmax, dt=0.1, exit_range=EXIT_RANGE, accel=ACCEL, lane_len=LANE_START - 25.0,
                 light_cycle=0, seed=99)
    trace, n_fin = [], 0
    for t in range(steps):
        raw = rng.uniform(-1.0, 1.0, (B, 2)).astype(np.float32)
        flow = dict(const, active=active, timer=timer, emitted=emitted, sim_step=sim_step, counter=t + 1)
        # the step with the flow rule (flow_rule_case holds it to step + eb_traffic_flow_step), then the reset's calls
        a = m.env_step(tr, obs, raw, ego, cand, mode, ref_idx=ref, cand_lw=lw, v_light=light, virtual=virtual, flow=flow)
        sc, o5, d16, ego1, par1, cand1, obs1, done1, act1, tim1, emi1, sim1, mode1, light1 = a
        mask = (done1 != 0).astype(np.uint8)
        ego2, par2, ref2, vnext, _ = m.env_reset(B, 777, t + 1, 1, ego1, par1, ref, mask=mask)
        cand2, act2, tim2, emi2, sim2, ph2, mode2, light2 = tr.traffic_flow_reset(K, mask, ego2, cand1, act1, tim1, emi1, sim1, phase0, lane, period,
                                                                                  vmax, clen, LANE_START - 25.0, rp, 1, 4242, t + 1, mode1, light1)
        obs2 = m.get_obs(ego2, cand2, mode2, light2, ref_idx=ref2, virtual=virtual, row_mask=mask, obs_init=obs1)
        virt2 = np.where(mask != 0, vnext, virtual).astype(np.uint8)
        final = np.where(mask[:, None] != 0, obs1, np.float32(np.nan))
        # the one call
        g = m.env_step(tr, obs, raw, ego, cand, mode, ref_idx=ref, cand_lw=lw, v_light=light, virtual=virtual, flow=flow,
                       auto_reset=dict(seed=777, counter=t + 1, training=1,
                                       flow=dict(cand_len=clen, phase0=phase0, random_phase=rp, seed=4242, counter=t + 1)))
        want = [sc, o5, d16, ego2, par2, cand2, obs2, done1, ref2, virt2, light2, final, ph2, act2, tim2, emi2, sim2, mode2, light2]
        names = ['scaled', 'out5', 'dict16', 'ego', 'params', 'cand', 'obs', 'done', 'ref_idx', 'virtual', 'v_light', 'final_obs', 'phase0',
                 'active', 'timer', 'emitted', 'sim_step', 'cand_mode', 'v_light (flow)']
        assert len(g) == len(want)
        for k, (x, y) in enumerate(zip(g, want)):
            if names[k] == 'cand':       # a vacant slot keeps whatever record it held: compare the occupied ones and the vacated ones' stale records alike
                assert np.array_equal(np.asarray(x).reshape(y.shape), y), (t, names[k])
                continue
            assert np.array_equal(np.asarray(x).reshape(np.asarray(y).shape), y, equal_nan=True), (t, names[k])
        n_fin += int(mask.sum())
        ego, cand, obs, ref, virtual, phase0 = g[3], g[5], g[6], g[8], g[9], g[12]
        active, timer, emitted, sim_step, mode, light = g[13], g[14], g[15], g[16], g[17], g[18]
        trace.append([np.asarray(x) for x in g])
    assert not strict or n_fin > B // 8, n_fin
    return trace
