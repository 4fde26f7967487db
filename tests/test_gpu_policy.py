"""GPU (-m gpu): the fused MLP kernel (env_build_amd/csrc/eb_policy.hip, fp32 matrix cores) and the shield loop
against the CPU oracle through the C-ABI — BIT-EXACT: v_mfma_f32_32x32x2_f32 accumulates as the fmaf chain the
oracle spells out, exp / tanh are the same deterministic kernels on both sides."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from env_build_amd.synthetic import assemble_obs, make_rollout_inputs  # noqa: E402
from tests._helpers import DeviceModel, HostModel, oracle_lib  # noqa: E402
from tests.test_policy_oracle import make_layers, torch_mlp  # noqa: E402

pytestmark = pytest.mark.gpu
PEN_RTOL = 1e-6

CONFIGS = [
    # obs_dim, n_hidden, n_units, out_dim, hidden act, out act
    (41, 2, 256, 4, 'elu', 'linear'), (137, 2, 256, 4, 'elu', 'linear'), (29, 1, 64, 4, 'relu', 'linear'),
    (45, 3, 128, 1, 'tanh', 'relu'), (265, 2, 512, 4, 'elu', 'tanh'), (33, 4, 100, 6, 'elu', 'linear'),
    (8, 8, 32, 2, 'elu', 'linear'), (137, 2, 300, 32, 'relu', 'linear'), (300, 1, 256, 4, 'elu', 'linear'),
]


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize('cfg', CONFIGS, ids=lambda c: '%dx%dx%d_%s' % (c[0], c[1], c[2], c[4]))
def test_mlp_kernel_equals_oracle_bitwise(cfg):
    obs_dim, n_hidden, n_units, out_dim, hact, oact = cfg
    rng = np.random.default_rng(obs_dim * 7 + n_units)
    host, dev = HostModel(oracle_lib(), 'left'), DeviceModel('left')
    layers = make_layers(rng, obs_dim, n_hidden, n_units, out_dim)
    scale = rng.uniform(0.05, 1.0, obs_dim).astype(np.float32)
    for sc in (None, scale):
        mh = host.make_mlp(obs_dim, n_hidden, n_units, out_dim, hact, oact, layers, sc)
        md = dev.make_mlp(obs_dim, n_hidden, n_units, out_dim, hact, oact, layers, sc)
        for n in (1, 63, 64, 65, 1000):
            obs = (rng.standard_normal((n, obs_dim)) * 3).astype(np.float32)
            want, got = host.mlp_forward(mh, out_dim, obs), dev.mlp_forward(md, out_dim, obs)
            assert same(got, want), 'n=%d: %d of %d logits differ, max |d| %.3g' % (
                n, int((got != want).sum()), got.size, float(np.max(np.abs(got - want))))
            if out_dim % 2 == 0:
                for rng_a in (1.0, 0.5, -1.0):
                    assert same(dev.policy_run_batch(md, out_dim // 2, obs, rng_a), host.policy_run_batch(mh, out_dim // 2, obs, rng_a))
        # and the torch fp32 restatement within the stated tolerance
        ref = torch_mlp(layers, obs, hact, oact, sc)
        assert np.max(np.abs(got - ref)) <= 1e-5 * max(1.0, float(np.max(np.abs(ref))))
        host.api.mlp_destroy(mh); dev.api.mlp_destroy(md)


def test_mlp_kernel_special_values_and_wide_range():
    """Large / tiny / non-finite inputs take the same path through exp, tanh and the chain on both sides."""
    rng = np.random.default_rng(5)
    host, dev = HostModel(oracle_lib(), 'left'), DeviceModel('left')
    for hact in ('elu', 'tanh', 'relu'):
        layers = make_layers(rng, 16, 2, 64, 4, bias_scale=1.0)
        mh = host.make_mlp(16, 2, 64, 4, hact, 'linear', layers)
        md = dev.make_mlp(16, 2, 64, 4, hact, 'linear', layers)
        obs = (rng.standard_normal((512, 16)) * np.exp(rng.uniform(-30, 12, (512, 1)))).astype(np.float32)
        obs[:8, 3] = [np.inf, -np.inf, np.nan, 0.0, -0.0, 1e-38, -1e-38, 3e38]
        assert same(dev.mlp_forward(md, 4, obs), host.mlp_forward(mh, 4, obs))
        assert same(dev.policy_run_batch(md, 2, obs, 1.0), host.policy_run_batch(mh, 2, obs, 1.0))
        host.api.mlp_destroy(mh); dev.api.mlp_destroy(md)


def test_policy_at_headline_size_sampled_rows():
    """65 536 observations of configs[2] width (D = 137): every 97th row against the oracle, all rows finite."""
    rng = np.random.default_rng(11)
    B, D = 65536, 137
    host, dev = HostModel(oracle_lib(), 'left'), DeviceModel('left')
    layers = make_layers(rng, D, 2, 256, 4)
    scale = rng.uniform(0.02, 0.2, D).astype(np.float32)
    mh = host.make_mlp(D, 2, 256, 4, 'elu', 'linear', layers, scale)
    md = dev.make_mlp(D, 2, 256, 4, 'elu', 'linear', layers, scale)
    obs = (rng.standard_normal((B, D)) * 10).astype(np.float32)
    got = dev.policy_run_batch(md, 2, obs, 1.0)
    assert np.all(np.isfinite(got)) and np.all(np.abs(got) <= 1.0)
    rows = np.arange(0, B, 97)
    assert same(got[rows], host.policy_run_batch(mh, 2, obs[rows], 1.0))
    host.api.mlp_destroy(mh); dev.api.mlp_destroy(md)


@pytest.mark.parametrize('task,N', [('left', 8), ('straight', 9), ('right', 5), ('left', 32)])
def test_shield_equals_oracle_bitwise(task, N):
    B = 1000
    host, dev = HostModel(oracle_lib(), task, n_veh=N), DeviceModel(task, n_veh=N)
    inp = make_rollout_inputs(task, B, N, 5, seed=21)
    trk = host.tracking_error(inp['ego'][:, 3], inp['ego'][:, 4], inp['ego'][:, 5], inp['ego'][:, 0], 0, ref_idx=inp['ref_idx'])
    obs0 = assemble_obs(inp['ego'], trk, inp['veh'])
    rng = np.random.default_rng(N)
    layers = make_layers(rng, host.D, 2, 256, 4)
    scale = rng.uniform(0.02, 0.2, host.D).astype(np.float32)
    mh = host.make_mlp(host.D, 2, 256, 4, 'elu', 'linear', layers, scale)
    md = dev.make_mlp(host.D, 2, 256, 4, 'elu', 'linear', layers, scale)
    for penalty, steps in ((0, 5), (1, 20), (0, 1), (1, 2)):
        want = host.shield_is_safe(mh, obs0, ref_idx=inp['ref_idx'], steps=steps, penalty=penalty)
        got = dev.shield_is_safe(md, obs0, ref_idx=inp['ref_idx'], steps=steps, penalty=penalty)
        for name, g, w in zip(('safe', 'punish', 'last obs', 'last actions'), got, want):
            if name == 'punish':    # sums of the rollout kernel's penalty outputs: rtol 1e-6 (tests/test_gpu_parity.py)
                np.testing.assert_allclose(g, w, rtol=PEN_RTOL * steps, atol=0)
            else:
                assert same(g, w), '%s differs (penalty %d, %d steps)' % (name, penalty, steps)
    assert 0 < int(want[0].sum()) < B
    host.api.mlp_destroy(mh); dev.api.mlp_destroy(md)


def test_facade_policy_and_native_shield():
    """LoadPolicy / Policy4Toyota / MLPNet (env_build_amd/policy.py) and shield.is_safe with the native policy ==
    the generic loop driven through the same policy as a plain callable."""
    import torch
    from types import SimpleNamespace
    from env_build_amd.dynamics_and_models import EnvironmentModel
    from env_build_amd.policy import LoadPolicy, MLPNet
    from env_build_amd.shield import is_safe, safe_shield
    task, N, B = 'left', 8, 777
    model = EnvironmentModel(task, 0, mode='selecting', n_veh=N)
    D = model.obs_dim
    args = SimpleNamespace(obs_dim=D, act_dim=2, num_hidden_layers=2, num_hidden_units=256, hidden_activation='elu',
                           policy_out_activation='linear', action_range=1.0, deterministic_policy=True,
                           obs_preprocess_type='scale', obs_scale=[0.2] * 6 + [1., 1 / 30., 0.2] + [1 / 30., 1 / 30., 0.2, 1 / 180.] * N)
    pol = LoadPolicy(args=args)
    inp = make_rollout_inputs(task, B, N, 5, seed=2)
    host = HostModel(oracle_lib(), task, n_veh=N, mode='selecting')
    trk = host.tracking_error(inp['ego'][:, 3], inp['ego'][:, 4], inp['ego'][:, 5], inp['ego'][:, 0], 0, path_id=1)
    obs0 = assemble_obs(inp['ego'], trk, inp['veh'])
    # facade output == oracle with the same weights
    w = pol.policy.policy.get_weights()
    mh = host.make_mlp(D, 2, 256, 4, 'elu', 'linear', list(zip(w[0::2], w[1::2])), np.asarray(args.obs_scale, np.float32))
    assert same(pol.run_batch(obs0).numpy(), host.policy_run_batch(mh, 2, obs0, 1.0))
    v = pol.obj_value_batch(obs0).numpy()
    assert v.shape == (B,) and np.all(v >= 0)
    # native shield (one C call) == generic loop with the policy as an opaque callable
    safe_n, pun_n = is_safe(model, pol, obs0, path_index=1, steps=5)
    last_n = model.obses.numpy()
    safe_g, pun_g = is_safe(model, lambda o: pol.run_batch(o), obs0, path_index=1, steps=5)
    assert same(safe_n.numpy(), safe_g.numpy()) and same(pun_n.numpy(), pun_g.numpy()) and same(last_n, model.obses.numpy())   # same kernels either way
    want = host.shield_is_safe(mh, obs0, path_id=1, steps=5, penalty=0)
    assert same(safe_n.numpy().astype(np.uint8), want[0])
    np.testing.assert_allclose(pun_n.numpy(), want[1], rtol=5 * PEN_RTOL, atol=0)
    act, started = safe_shield(model, pol, obs0, path_index=1)
    a = act.numpy()
    assert same(started.numpy(), ~safe_n.numpy()) and np.all(a[started.numpy()] == np.array([0., -1.], np.float32))
    # weights round trip and a second network shape
    net = MLPNet(D, 3, 100, 'tanh', 1, name='obj_v', output_activation='relu', seed=3)
    w2 = net.get_weights()
    net.set_weights([x * 0.5 for x in w2])
    assert not same(net(obs0).numpy(), MLPNet(D, 3, 100, 'tanh', 1, name='obj_v', output_activation='relu', seed=3)(obs0).numpy())
    with pytest.raises(ValueError):
        net(np.zeros((3, D + 1), np.float32))
    torch.cuda.synchronize()


def test_policy_and_traffic_entry_points_reject_bad_arguments():
    """Error behaviour of the newer entry points on the HIP library: bad handles, sizes and pointers come back as
    ValueError / EbError with a message, nothing is launched and nothing crashes; the same calls fail the same way
    in the oracle."""
    import ctypes as C
    from env_build_amd import _capi
    rng = np.random.default_rng(0)
    for mdl in (HostModel(oracle_lib(), 'left'), DeviceModel('left')):
        api = mdl.api
        with pytest.raises(ValueError):
            mdl.make_mlp(8, 9, 64, 4, 'elu', 'linear', [])                      # too many hidden layers
        with pytest.raises(ValueError):
            mdl.make_mlp(8, 1, 64, 4, 'elu', 'linear', make_layers(rng, 8, 1, 64, 4)[:1])   # a layer missing
        cfg = _capi.EbMlpConfig(_capi.EB_ABI_VERSION + 1, 8, 1, 64, 4, 2, 0, 0)
        h = C.c_void_p()
        with pytest.raises(ValueError):
            api.mlp_create(C.byref(cfg), C.byref(h))                             # ABI version mismatch
        cfg = _capi.EbMlpConfig(_capi.EB_ABI_VERSION, 8, 1, 64, 4, 7, 0, 0)
        with pytest.raises(ValueError):
            api.mlp_create(C.byref(cfg), C.byref(h))                             # unknown activation
        cfg = _capi.EbMlpConfig(_capi.EB_ABI_VERSION, 8, 1, 64, 4, 2, 0, 0)
        api.mlp_create(C.byref(cfg), C.byref(h))
        obs = mdl._in(np.zeros((4, 8), np.float32))
        out = mdl._out((4, 4))
        with pytest.raises(_capi.EbError):
            api.mlp_forward(h, 4, mdl._ptr(obs), mdl._ptr(out), mdl.stream)      # weights never set: EB_ESTATE
        with pytest.raises(ValueError):
            api.mlp_set_layer(h, 5, mdl._ptr(obs), None)                         # layer out of range, null bias
        with pytest.raises(ValueError):
            api.mlp_forward(None, 4, mdl._ptr(obs), mdl._ptr(out), mdl.stream)
        api.mlp_destroy(h)
        assert api.lib.eb_mlp_destroy(None) == 0
        # shield: policy that does not fit the model, bad step count, aliased work buffers
        good = mdl.make_mlp(mdl.D, 1, 64, 4, 'elu', 'linear', make_layers(rng, mdl.D, 1, 64, 4))
        ob = np.zeros((4, mdl.D), np.float32)
        ri = np.zeros(4, np.int32)
        with pytest.raises(ValueError):
            mdl.shield_is_safe(good, ob, ref_idx=ri, steps=0)
        with pytest.raises(ValueError):
            mdl.shield_is_safe(good, ob, ref_idx=ri, penalty=5)
        with pytest.raises(ValueError):
            mdl.shield_is_safe(None, ob, ref_idx=ri)
        a = mdl._in(ob); o5 = mdl._out((5, 4)); act = mdl._out((4, 2)); pu = mdl._out((4,)); sf = mdl._out((4,), np.uint8)
        with pytest.raises(ValueError):                                           # obs_a == obs_b
            api.shield_is_safe(mdl.h, good, 4, mdl._ptr(mdl._in(ob)), mdl._ptr(mdl._in(ri, np.int32)), 0, 5, 0, C.c_float(1.0),
                               mdl._ptr(a), mdl._ptr(a), mdl._ptr(act), mdl._ptr(o5), mdl._ptr(pu), mdl._ptr(sf), mdl.stream)
        api.mlp_destroy(good)
        # traffic kernels
        c = mdl._in(np.zeros((2, 4, 4), np.float32)); en = mdl._in(np.zeros((4, 5), np.float32))
        with pytest.raises(ValueError):
            api.traffic_respawn(mdl.h, 2, 0, mdl._ptr(c), mdl._ptr(en), C.c_float(65.), C.c_float(60.), C.c_float(8.),
                                C.c_uint64(1), C.c_uint64(1), None, None, None, C.c_float(0.), mdl.stream)  # m_cand < 1
        with pytest.raises(ValueError):
            api.traffic_respawn(mdl.h, 2, 4, None, mdl._ptr(en), C.c_float(65.), C.c_float(60.), C.c_float(8.),
                                C.c_uint64(1), C.c_uint64(1), None, None, None, C.c_float(0.), mdl.stream)
        with pytest.raises(ValueError):
            api.traffic_flow_step(mdl.h, 2, 6, None, None, None, None, None, None, None, None, C.c_float(0.1), C.c_float(65.),
                                  C.c_float(2.6), C.c_float(75.), 1, C.c_uint64(1), C.c_uint64(1), None, None, mdl.stream)   # 72 slots
        assert api.lib.eb_traffic_respawn(mdl.h, 0, 4, None, None, C.c_float(1), C.c_float(1), C.c_float(1), 1, 1, None, None, None, C.c_float(0.), None) == 0


# ---- G13: fixtures from the reference's own MLPNet / Policy4Toyota / Preprocessor / LoadPolicy.run_batch ----
from tests._helpers import close, golden  # noqa: E402
from tests.test_policy_oracle import G13, g13_layers  # noqa: E402


@pytest.mark.parametrize('name', G13)
def test_g13_policy_fixtures_on_gpu_through_the_kernel_and_the_facade(name):
    """the MLP kernel behind the C-ABI, and the drop-in classes (env_build_amd.policy.LoadPolicy with the reference's
    set_weights list: [obj_v weights, policy weights], kernel then bias per layer) against the reference's outputs"""
    from types import SimpleNamespace
    import torch
    from env_build_amd.policy import LoadPolicy
    g = golden(name)
    obs, scale, hidden, units, act = g['obs'], g['obs_scale'], int(g['hidden']), int(g['units']), str(g['act'])
    dev = DeviceModel('left')
    pol = dev.make_mlp(obs.shape[1], hidden, units, 4, act, 'linear', g13_layers(g, 'policy'), scale)
    val = dev.make_mlp(obs.shape[1], hidden, units, 1, act, 'relu', g13_layers(g, 'obj_v'), scale)
    close(dev.policy_run_batch(pol, 2, obs, 1.0), g['actions'], 1e-5, 5e-6, 'GPU G13 policy actions')
    close(dev.mlp_forward(val, 1, obs)[:, 0], g['values'], 1e-5, 5e-6, 'GPU G13 obj_v values')
    dev.api.mlp_destroy(pol); dev.api.mlp_destroy(val)
    args = SimpleNamespace(obs_dim=obs.shape[1], act_dim=2, num_hidden_layers=hidden, num_hidden_units=units,
                           hidden_activation=act, policy_out_activation='linear', action_range=1.0, deterministic_policy=True,
                           obs_preprocess_type='scale', obs_scale=[float(x) for x in scale])
    lp = LoadPolicy(args=args, device=torch.device('cuda', 0))
    n = 2 * (hidden + 1)
    lp.policy.set_weights([[g['obj_v_w%d' % i] for i in range(n)], [g['policy_w%d' % i] for i in range(n)]])
    close(lp.run_batch(obs).numpy(), g['actions'], 1e-5, 5e-6, 'GPU G13 facade run_batch')
    close(lp.obj_value_batch(obs).numpy(), g['values'], 1e-5, 5e-6, 'GPU G13 facade obj_value_batch')


from tests.test_policy_oracle import G14, g14_check  # noqa: E402


@pytest.mark.parametrize('name', G14)
def test_g14_shield_fixtures_on_gpu_through_the_kernels_and_the_facade(name):
    """eb_shield_is_safe (policy kernel -> rollout kernel -> accumulate, 5 steps) and the drop-in safe_shield of
    env_build_amd.shield against HierarchicalDecision.is_safe / safe_shield of the reference (fixture G14)"""
    from types import SimpleNamespace
    import torch
    from env_build_amd.dynamics_and_models import EnvironmentModel
    from env_build_amd.policy import LoadPolicy
    from env_build_amd.shield import safe_shield
    g = golden(name)
    task = name.split('_')[-1]
    n = len([k for k in g.files if k.startswith('policy_w')])
    layers = [(g['policy_w%d' % (2 * i)], g['policy_w%d' % (2 * i + 1)]) for i in range(n // 2)]
    hidden, units, D = n // 2 - 1, layers[0][0].shape[1], g['obs'].shape[1]
    dev = DeviceModel(task, mode='selecting')
    mlp = dev.make_mlp(D, hidden, units, 4, 'elu', 'linear', layers, g['obs_scale'])
    g14_check(dev, g, mlp)
    dev.api.mlp_destroy(mlp)
    # the classes a user of the reference would switch to
    cuda = torch.device('cuda', 0)
    args = SimpleNamespace(obs_dim=D, act_dim=2, num_hidden_layers=hidden, num_hidden_units=units, hidden_activation='elu',
                           policy_out_activation='linear', action_range=1.0, deterministic_policy=True,
                           obs_preprocess_type='scale', obs_scale=[float(x) for x in g['obs_scale']])
    lp = LoadPolicy(args=args, device=cuda)
    lp.policy.policy.set_weights([g['policy_w%d' % i] for i in range(n)])
    model = EnvironmentModel(task, num_future_data=0, mode='selecting', device=cuda)
    act, started = safe_shield(model, lp, g['obs'], path_index=int(g['path_index']))
    assert np.array_equal(started.numpy().astype(np.uint8), g['shield_started'])
    close(act.numpy(), g['safe_action'], 1e-5, 5e-6, 'GPU G14 facade safe_shield actions')
