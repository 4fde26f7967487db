"""Fixture checks shared by the CPU suite (oracle vs fixture, tests/test_oracle_golden.py) and the GPU suite (HIP
library vs the SAME fixtures, tests/test_gpu_golden.py): each takes `make(task, **kw)` -> a HostModel / DeviceModel and
a label prefix for the parity-margin table.  The fixtures hold the REFERENCE's outputs (oracle/gen_golden.py)."""
import numpy as np

from tests._helpers import close, golden

RTOL = 1e-5
# absolute slack next to north_star's rtol 1e-5, per check — set from the OBSERVED excess (printed at the end of a run
# under "parity margins"), not the other way round.  1 step = the transcendental kernels' <= 2 ulp; closed loops compound it.
ATOL = dict(g2=1e-6, g2_cancel=5e-6, g3=5e-6, g4=5e-6, g5_teacher=5e-6, g5_closed_loop=5e-6, g7_reward=5e-6, g7_state=5e-6,
            g8=5e-6, g9=5e-6, g12=5e-6)
REWARD_KEYS = ('punish_steer', 'punish_a_x', 'punish_yaw_rate', 'devi_v', 'devi_y', 'devi_phi',
               'scaled_punish_steer', 'scaled_punish_a_x', 'scaled_punish_yaw_rate', 'scaled_devi_v',
               'scaled_devi_y', 'scaled_devi_phi', 'veh2veh4training', 'veh2road4training', 'veh2veh4real',
               'veh2road4real')  # DAM:302-318


def check_g2_f_xu(make, tag=''):
    """G2 (DAM:52-83): 256 states incl. v_x = 0, a_x sign changes, edge headings; two step sizes."""
    g = golden('g2_f_xu')
    m = make('left')
    for name, tau in zip(('tau0p1', 'tau0p05'), g['taus']):
        nxt, par = m.f_xu(g['states'], g['actions'], float(tau))
        # v_x, x, y: no cancellation -> tight; v_y, r: sums of 1e4..1e5-magnitude terms that cancel
        close(nxt[:, [0, 3, 4, 5]], g['next_' + name][:, [0, 3, 4, 5]], RTOL, ATOL['g2'], tag + 'G2 f_xu next (v_x, x, y, phi)')
        close(nxt[:, 1:3], g['next_' + name][:, 1:3], RTOL, ATOL['g2_cancel'], tag + 'G2 f_xu next (v_y, r)')
        close(par, g['params_' + name], RTOL, ATOL['g2'], tag + 'G2 f_xu params')


def check_g3_compute_rewards(make, task, tag=''):
    """G3 (DAM:186-320): vehicles at controlled distances around 2.5 / 3.5 m, ego points around every road wall."""
    g = golden('g3_rewards_%s' % task)
    m = make(task)
    out5, d16 = m.compute_rewards(g['obs'], g['actions'])
    assert [str(k) for k in g['dict_keys']] == list(REWARD_KEYS)
    close(out5, g['out5'], RTOL, ATOL['g3'], tag + 'G3 compute_rewards out5 (%s)' % task)
    close(d16, g['dict16'], RTOL, ATOL['g3'], tag + 'G3 compute_rewards dict16 (%s)' % task)
    # the penalty MASKS (which envs are penalised at all) are bit-exact
    assert np.array_equal(out5[1:] > 0, g['out5'][1:] > 0)
    assert np.array_equal(d16[12:] > 0, g['dict16'][12:] > 0)


def check_g4_reference_own_vector(make, tag=''):
    """The reference's only known-input vector (DAM:803-811, task 'straight', n = 10)."""
    g = golden('g4_tracking')
    m = make('straight')
    for k in range(3):
        out = m.tracking_error(g['ref_xs'], g['ref_ys'], g['ref_phis'], g['ref_vs'], 10, path_id=k)
        close(out, g['ref_out_path%d_n10' % k], RTOL, ATOL['g4'], tag + 'G4 reference own vector (DAM:805-808)')


def check_g4_tracking(make, task, tag=''):
    g = golden('g4_tracking')
    m = make(task)
    for k in range(3):
        t = '%s_p%d' % (task, k)
        x, y, phi, v = g['x_' + t], g['y_' + t], g['phi_' + t], g['v_' + t]
        idx, pts = m.find_closest_point(x, y, path_id=k)
        assert np.array_equal(idx.astype(np.int64), g['index_' + t])        # argmin: bit-exact
        assert np.array_equal(pts, g['points_' + t])                         # gather: bit-exact
        for nf in (0, 3):
            out = m.tracking_error(x, y, phi, v, nf, path_id=k)
            ref = g['out_%s_n%d' % (t, nf)]
            close(out, ref, RTOL, ATOL['g4'], tag + 'G4 tracking_error_vector (%s)' % task)
            assert np.array_equal(out[:, 2], ref[:, 2])                      # v - 8: exact


def check_g5t_teacher_forced_n32(make, task, tag=''):
    g = golden('g5t_teacher_%s_N32' % task)
    m = make(task, n_veh=32, mode='training', modes=[str(s) for s in g['modes']])
    for t in range(g['actions'].shape[0]):
        obs, o5, _ = m.rollout_step(g['obs_all'][t], g['actions'][t], g['ref_idx'])
        close(obs, g['obs_all'][t + 1], RTOL, ATOL['g5_teacher'], tag + 'G5T teacher-forced: obs (N = 32)')
        close(o5, g['out5'][t], RTOL, ATOL['g5_teacher'], tag + 'G5T teacher-forced: out5 (N = 32)')
        assert np.array_equal(o5[1:] > 0, g['out5'][t][1:] > 0) or _only_threshold_flips(o5, g['out5'][t])


def _only_threshold_flips(o5, ref):
    """penalty masks may differ only where the penalty itself is below the absolute tolerance (a distance within ~1e-6 m
    of a threshold)"""
    flip = (o5[1:] > 0) != (ref[1:] > 0)
    return bool(np.all(np.maximum(np.abs(o5[1:][flip]), np.abs(ref[1:][flip])) < ATOL['g5_teacher']))


def check_g9_ss(make, task, tag=''):
    g = golden('g9_ss_%s' % task)
    m = make(task)
    out = m.ss(g['obs'], g['actions'], g['ref_idx'], 0, float(g['lam']))
    close(out, g['out'], RTOL, ATOL['g9'], tag + 'G9 ss (%s)' % task)
    assert np.array_equal(out > 0, g['out'] > 0)


def check_g12_exit_frames(make, tag=''):
    """G12 (UTL:184-196 cal_ego_info_in_transform_coordination, generated from the reference in float64): the rows whose
    rotation is one of the four exit angles go through eb_exit_frame — shifted on the host (the kernel's frames have no
    shift), rotated in fp32 on the device, and back again with the inverse flag."""
    g = golden('g12_utl_frames')
    m = make('left')
    rot = g['rotate']
    rows = np.array([i for i in range(len(rot)) if float(rot[i]) in (0.0, 90.0, 180.0, -90.0)])
    assert len(rows) >= 24
    exit_id = np.array([{0.0: 0, 90.0: 1, 180.0: 2, -90.0: 3}[float(rot[i])] for i in rows], np.uint8)
    assert sorted(set(exit_id.tolist())) == [0, 1, 2, 3]
    ego = np.zeros((len(rows), 6), np.float32)
    ego[:, 0:3] = (3.0, 0.25, -0.125)
    ego[:, 3] = g['x'][rows] - g['shift_x'][rows]
    ego[:, 4] = g['y'][rows] - g['shift_y'][rows]
    ego[:, 5] = g['d'][rows]
    out = m.exit_frame(exit_id, ego)
    assert np.array_equal(out[:, :3], ego[:, :3])
    close(out[:, 3:5], g['out_ego'][rows][:, :2], RTOL, ATOL['g12'], tag + 'G12 exit frame: ego x, y (UTL:184-196)')
    want_phi = g['out_ego'][rows][:, 2]
    d = (out[:, 5].astype(np.float64) - want_phi + 180.0) % 360.0 - 180.0      # -180 and 180 are the same heading
    close(want_phi + d, want_phi, RTOL, 2e-5, tag + 'G12 exit frame: ego phi (UTL:184-196)')
    back = m.exit_frame(exit_id, out, inverse=True)
    close(back[:, 3:5], ego[:, 3:5], RTOL, ATOL['g12'], tag + 'G12 exit frame: there and back')


def check_g7_through_env_step(make, tag=''):
    """G7 = BASELINE.json configs[0]: one `left` env, 8 surrounding vehicles, 200 steps (SUMO-free composition generated from
    the reference's own methods) — through eb_env_step, the composite entry CrossroadEnd2end.step uses (one launch on the
    GPU), with the state carried from step to step by the library under test."""
    from env_build_amd import _capi
    from env_build_amd.endtoend_env_utils import VEHICLE_MODE_LIST
    g = golden('g7_config1_left')
    modes = [str(m) for m in g['modes']]
    assert modes == VEHICLE_MODE_LIST['left']
    m, tr = make('left', mode='selecting'), make('left', n_veh=len(modes), modes=modes)
    H = g['actions'].shape[0]
    ref = np.array([int(g['ref_index'])], np.int32)
    ego, veh, obs = g['ego'][0:1].copy(), g['veh'][0][None].copy(), g['obs'][0:1].copy()
    cmode = np.array([[_capi.VMODE_ID[x] for x in modes]], np.uint8)
    n_done_mismatch = 0
    for t in range(H):
        _, out5, _, ego, _, veh, obs, done = m.env_step(tr, obs, g['actions'][t:t + 1], ego, veh, cmode, ref_idx=ref,
                                                        v_light=np.zeros(1, np.uint8))
        close(out5[0, 0], g['reward'][t], RTOL, ATOL['g7_reward'], tag + 'G7 through eb_env_step: reward')
        close(ego[0], g['ego'][t + 1], RTOL, ATOL['g7_state'], tag + 'G7 through eb_env_step: ego')
        close(veh[0], g['veh'][t + 1], RTOL, ATOL['g7_state'], tag + 'G7 through eb_env_step: vehicles')
        close(obs[0], g['obs'][t + 1], RTOL, ATOL['g7_state'], tag + 'G7 through eb_env_step: obs')
        n_done_mismatch += int(done[0] != g['done_code'][t])
    assert n_done_mismatch == 0
