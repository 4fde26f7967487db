"""CPU (-m "not gpu"): the C-ABI libraries load and export every symbol include/envbuild.h
declares (no compute calls on the HIP library without a GPU), the ctypes prototypes cover the
header, argument validation / error reporting, and the host-side helpers."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from env_build_amd import _capi, build as eb_build
from env_build_amd import endtoend_env_utils as U
from env_build_amd.synthetic import assemble_obs, make_rollout_inputs
from tests._helpers import golden, ROOT, HostModel, oracle_lib, _p

HEADER = os.path.join(ROOT, 'include', 'envbuild.h')


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(eb_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_what_ctypes_binds():
    syms = header_symbols()
    assert len(syms) >= 20
    assert sorted(_capi.PROTOTYPES) == syms


def test_hip_library_builds_loads_and_exports_every_symbol():
    lib_path = eb_build.build()            # hipcc --offload-arch=gfx950 (cross-compiles without a GPU)
    assert os.path.isfile(lib_path)
    import torch  # noqa: F401  (binds the HIP runtime torch ships before ours, as the product does)
    lib = C.CDLL(lib_path)
    for name in header_symbols():
        assert hasattr(lib, name), name
    lib.eb_backend.restype = C.c_char_p
    assert lib.eb_backend() == b'hip'
    assert lib.eb_abi_version() == _capi.EB_ABI_VERSION


def test_hip_library_contains_gfx950_code_object():
    lib_path = eb_build.build()
    blob = open(lib_path, 'rb').read()
    assert b'gfx950' in blob and b'rollout_fused_4x8' in blob


def test_hip_backend_refuses_to_run_without_a_gpu():
    """No CPU fallback: on a box without a GPU eb_create fails loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible')
    api = _capi.hip_api()
    with pytest.raises((_capi.EbError, ValueError)) as e:
        api.create('left', 8, 0, _capi.MODE_TRAINING)
    assert 'no HIP device' in str(e.value) or 'HIP' in str(e.value)
    from env_build_amd.dynamics_and_models import EnvironmentModel
    with pytest.raises(_capi.EbError):
        EnvironmentModel('left')


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'env_build_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'libenvbuild_oracle' not in text and 'oracle_lib' not in text, f
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f


def test_oracle_exports_and_error_paths():
    api = oracle_lib()
    for name in header_symbols():
        assert hasattr(api.lib, name), name
    with pytest.raises(ValueError):
        api.create('left', 0, 0, _capi.MODE_TRAINING)          # n_veh out of range
    with pytest.raises(ValueError):
        api.create('left', 65, 0, _capi.MODE_TRAINING)
    with pytest.raises(ValueError):
        api.create(7, 8, 0, _capi.MODE_TRAINING)               # bad task
    h = api.create('left', 8, 0, _capi.MODE_TRAINING)
    z = np.zeros((4, 41), np.float32)
    a = np.zeros((4, 2), np.float32)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    with pytest.raises(_capi.EbError):                          # paths / modes not set -> EB_ESTATE
        api.rollout_step(h, 4, p(z), p(a), None, 0, p(z.copy()), p(np.zeros((5, 4), np.float32)), None, None)
    api.destroy(h)


def test_oracle_empty_and_bad_ref_index():
    host = HostModel(oracle_lib(), 'left')
    out, o5, sc = host.rollout_step(np.zeros((0, host.D), np.float32), np.zeros((0, 2), np.float32),
                                    np.zeros((0,), np.int32))
    assert out.shape == (0, host.D) and o5.shape == (5, 0)
    with pytest.raises(ValueError):    # training mode without ref_indexes
        host.rollout_step(np.zeros((4, host.D), np.float32), np.zeros((4, 2), np.float32), None)
    inp = make_rollout_inputs('left', 16, 8, 1, seed=0)
    inp['ref_idx'][::2] = 9            # rows with an out-of-range path keep zero tracking (DAM:342, 352)
    trk = host.tracking_error(inp['ego'][:, 3], inp['ego'][:, 4], inp['ego'][:, 5], inp['ego'][:, 0], 0,
                              ref_idx=inp['ref_idx'])
    assert np.all(trk[::2] == 0) and np.all(trk[1::2, 2] == inp['ego'][1::2, 0] - 8)


def test_oracle_linearity_free_properties():
    """Size-independent properties the domain offers: rewards do not depend on far-away vehicles;
    veh_predict keeps speed; a straight-mode vehicle keeps heading; tape == stepwise."""
    host = HostModel(oracle_lib(), 'straight', n_veh=16)
    inp = make_rollout_inputs('straight', 300, 16, 4, seed=2)
    trk = host.tracking_error(inp['ego'][:, 3], inp['ego'][:, 4], inp['ego'][:, 5], inp['ego'][:, 0], 0,
                              ref_idx=inp['ref_idx'])
    obs = assemble_obs(inp['ego'], trk, inp['veh'])
    act = host.action_transform(inp['actions'][0])
    o5, _ = host.compute_rewards(obs, act)
    far = obs.copy()
    veh = far[:, 9:].reshape(300, 16, 4)
    d = np.hypot(veh[:, :, 0] - far[:, None, 3], veh[:, :, 1] - far[:, None, 4])
    veh[d > 7.0, 0] += 500.0           # > 3.5 + 2*1.4 m away: contributes exact zeros
    o5b, _ = host.compute_rewards(far, act)
    assert np.array_equal(o5, o5b)
    nxt = host.veh_predict(inp['veh']).reshape(300, 16, 4)
    assert np.array_equal(nxt[:, :, 2], inp['veh'].reshape(300, 16, 4)[:, :, 2])
    modes = U.tiled_mode_list('straight', 16)
    straight_slots = [j for j, m in enumerate(modes) if m in ('du', 'ud')]
    np.testing.assert_allclose(nxt[:, straight_slots, 3], inp['veh'].reshape(300, 16, 4)[:, straight_slots, 3],
                               rtol=1e-6, atol=1e-4)
    a, o5s = host.rollout_tape(obs, inp['actions'], inp['ref_idx'])
    b = obs
    for t in range(4):
        b, o5t, _ = host.rollout_step(b, inp['actions'][t], inp['ref_idx'])
        assert np.array_equal(o5t, o5s[t])
    assert np.array_equal(a, b)


def test_constants_and_mode_tables():
    assert (U.L, U.W, U.LANE_WIDTH, U.LANE_NUMBER, U.CROSSROAD_SIZE, U.EXPECTED_V) == (4.8, 2.0, 3.75, 3, 50, 8.)
    assert U.VEH_NUM == dict(left=8, straight=9, right=5)
    assert U.VEHICLE_MODE_LIST['left'] == ['dl', 'dl', 'du', 'du', 'ud', 'ud', 'ul', 'ul']
    assert U.VEHICLE_MODE_LIST['right'] == ['dr', 'ur', 'ur', 'lr', 'lr']
    assert len(U.ROUTE2MODE) == 12 and all(U.ROUTE2MODE[U.MODE2ROUTE[m]] == m for m in _capi.VMODES)
    assert U.tiled_mode_list('right', 7) == ['dr', 'ur', 'ur', 'lr', 'lr', 'dr', 'ur']
    assert U.deal_with_phi(190) == -170 and U.deal_with_phi(-180) == 180 and U.deal_with_phi(540) == 180
    x, y, d = U.rotate_coordination(1.0, 0.0, 0.0, 90.0)
    assert abs(x) < 1e-12 and abs(y + 1.0) < 1e-12 and d == -90.0
    assert U.judge_feasible(1.0, -30.0, 'left') and not U.judge_feasible(4.0, -30.0, 'left')
    assert U.judge_feasible(5.0, 30.0, 'straight') and not U.judge_feasible(-1.0, 30.0, 'straight')
    assert U.judge_feasible(30.0, -5.0, 'right') and not U.judge_feasible(30.0, 1.0, 'right')
    assert U.judge_feasible(-24.0, 24.0, 'right')


def test_synthetic_inputs_are_seeded_and_shaped():
    a = make_rollout_inputs('left', 64, 32, 5, seed=3)
    b = make_rollout_inputs('left', 64, 32, 5, seed=3)
    c = make_rollout_inputs('left', 64, 32, 5, seed=4)
    for k in ('ego', 'veh', 'ref_idx', 'actions'):
        assert np.array_equal(a[k], b[k])
    assert not np.array_equal(a['ego'], c['ego'])
    assert a['ego'].shape == (64, 6) and a['veh'].shape == (64, 128) and a['actions'].shape == (5, 64, 2)
    assert a['ref_idx'].dtype == np.int32 and set(a['ref_idx'].tolist()) <= {0, 1, 2}
    assert a['modes'] == U.tiled_mode_list('left', 32)


def test_oracle_plan_summary_and_events():
    host = HostModel(oracle_lib(), 'left', n_veh=16)
    inp = make_rollout_inputs('left', 130, 16, 5, seed=9)
    trk = host.tracking_error(inp['ego'][:, 3], inp['ego'][:, 4], inp['ego'][:, 5], inp['ego'][:, 0], 0,
                              ref_idx=inp['ref_idx'])
    obs = assemble_obs(inp['ego'], trk, inp['veh'])
    out_a, o5_a = host.rollout_tape(obs, inp['actions'], inp['ref_idx'])
    out_b, o5_b, s8 = host.plan_run(obs, inp['actions'], inp['ref_idx'])
    assert np.array_equal(out_a, out_b) and np.array_equal(o5_a, o5_b)
    np.testing.assert_allclose(s8[0], o5_a[:, 0].astype(np.float64).sum(), rtol=1e-6)
    np.testing.assert_allclose(s8[1], o5_a[:, 1].astype(np.float64).sum(), rtol=1e-6)
    np.testing.assert_allclose(s8[2], o5_a[:, 2].astype(np.float64).sum(), rtol=1e-6)
    assert s8[3] == (o5_a[:, 2] > 0).any(0).sum()
    np.testing.assert_allclose(s8[4], np.abs(out_a[:, 6]).astype(np.float64).sum(), rtol=1e-6)
    assert s8[5] == np.abs(out_a[:, 6]).max() and s8[6] == 130 and s8[7] == 5
    api = host.api
    e0, e1, ms = C.c_void_p(), C.c_void_p(), C.c_float()
    api.event_create(host.h, C.byref(e0)); api.event_create(host.h, C.byref(e1))
    api.event_record(e0, None); api.event_record(e1, None)
    api.event_elapsed_ms(e0, e1, C.byref(ms))
    assert 0.0 <= ms.value < 1000.0
    api.event_destroy(e0); api.event_destroy(e1)
    with pytest.raises(ValueError):
        api.plan_create(host.h, 0, 5, None, None, None, 0, None, None, None, None, None, C.byref(C.c_void_p()))


def test_oracle_accumulating_rollout_equals_the_two_pass_summary():
    """ABI 5: eb_rollout_step_acc = eb_rollout_step + the episodic sums collected on the way; eb_episode_acc_finish gives
    eb_episode_summary's 8 floats; plans with a summary (own or caller's workspace) run the same thing."""
    host = HostModel(oracle_lib(), 'straight', n_veh=9)
    inp = make_rollout_inputs('straight', 77, 9, 6, seed=12)
    trk = host.tracking_error(inp['ego'][:, 3], inp['ego'][:, 4], inp['ego'][:, 5], inp['ego'][:, 0], 0, ref_idx=inp['ref_idx'])
    obs = assemble_obs(inp['ego'], trk, inp['veh'])
    out_a, o5_a = host.rollout_tape(obs, inp['actions'], inp['ref_idx'])
    want = host.episode_summary(o5_a, out_a)
    assert want[3] > 0                                               # the scene does punish somebody
    acc = host.acc_workspace(77, 6)
    for _ in range(2):                                               # the second rollout starts over in the used workspace
        out_b, o5_b, s8 = host.rollout_acc(obs, inp['actions'], inp['ref_idx'], acc=acc)
        assert np.array_equal(out_a, out_b) and np.array_equal(o5_a, o5_b)
        np.testing.assert_allclose(s8, want, rtol=1e-6, atol=0)
        assert s8[3] == want[3] and s8[5] == want[5] and s8[6] == 77 and s8[7] == 6
    for caller_acc in (True, 'finish'):
        out_c, o5_c, s8_c = host.plan_run(obs, inp['actions'], inp['ref_idx'], caller_acc=caller_acc)
        assert np.array_equal(out_a, out_c) and np.array_equal(o5_a, o5_c) and np.array_equal(s8_c, s8)
    api = host.api
    with pytest.raises(ValueError):
        api.rollout_step_acc(host.h, 77, _p(obs), _p(inp['actions'][0]), _p(inp['ref_idx']), 0, _p(out_a.copy()), _p(o5_a[0].copy()),
                             None, None, 0, 1, None, None)
    with pytest.raises(ValueError):                                  # a later step without the previous step's out5
        api.rollout_step_acc(host.h, 77, _p(obs), _p(inp['actions'][0]), _p(inp['ref_idx']), 0, _p(out_a.copy()), _p(o5_a[0].copy()),
                             None, _p(np.zeros(1 << 16, np.uint8)), 1, 6, None, None)
    nb = C.c_int64()
    api.episode_acc_bytes(host.h, 0, 6, C.byref(nb))
    assert nb.value == 64
    hip_nb = C.c_int64()            # the two libraries size the workspace alike (a caller may allocate once for either)
    api.episode_acc_bytes(host.h, 1000, 25, C.byref(nb))
    assert nb.value == ((1000 + 27) // 28) * (25 * 4 + 2) * 8 + 1000 + 64


def test_exit_frame_transforms_match_reference_values():
    """endtoend_env_utils' coordinate helpers (UTL:107-196) against fixture G12: the outputs of the reference's own
    functions on the same python floats (oracle/gen_golden.py g12) — equal to the last bit (same libm, same op order)."""
    from env_build_amd import endtoend_env_utils as U
    g = golden('g12_utl_frames')
    n = len(g['x'])
    for i in range(n):
        x, y, d = float(g['x'][i]), float(g['y'][i]), float(g['d'][i])
        sx, sy = float(g['shift_x'][i]), float(g['shift_y'][i])
        r = int(g['rotate'][i]) if i < n // 2 else float(g['rotate'][i])
        assert tuple(g['out_rotate'][i]) == U.rotate_coordination(x, y, d, r)
        assert tuple(g['out_shift_rotate'][i]) == U.shift_and_rotate_coordination(x, y, d, sx, sy, r)
        assert tuple(g['out_rotate_shift'][i]) == U.rotate_and_shift_coordination(x, y, d, sx, sy, r)
        t = U.cal_info_in_transform_coordination([dict(x=x, y=y, v=3.0, phi=d, w=2.0, l=4.8, route=('1o', '4i'))], sx, sy, r)[0]
        assert (t['x'], t['y'], t['phi']) == tuple(g['out_veh'][i]) and t['v'] == 3.0 and t['route'] == ('1o', '4i')
        ego = U.cal_ego_info_in_transform_coordination(
            dict(x=x, y=y, phi=d, Corner_point=[(x + 2.4, y + 1.0), (x - 2.4, y - 1.0)]), sx, sy, r)
        assert (ego['x'], ego['y'], ego['phi']) == tuple(g['out_ego'][i])
        assert np.array_equal(np.array(ego['Corner_point'], np.float64).ravel(), g['out_corners'][i])
