"""The policy side of the reference's decision loop, on the GPU: `MLPNet` (utils/model.py:18-43), `Policy4Toyota`
(utils/policy.py:19-101, the deterministic inference surface) and `LoadPolicy` (utils/load_policy.py:19-63) with
the 'scale' observation preprocessor (utils/preprocessor.py:116-123).

Same names, constructor arguments and methods as the reference; the arithmetic is one fused HIP kernel per call
(env_build_amd/csrc/eb_policy.hip: fp32 matrix cores, the whole network in one launch).  Training (optimisers,
stochastic sampling, log-probabilities) is out of scope — the shield and the path selection only ever call
`run_batch` / `obj_value_batch`.

TensorFlow checkpoints cannot be read here (no TF, and the reference's checkpoints are git-ignored): weights are
exchanged as the list `Model.get_weights()` returns — [kernel0, bias0, kernel1, bias1, ...], kernels [in, out] —
or as an .npz of that list (`save_weights` / `load_weights`).  There is no CPU path.
"""
import ctypes as C
import json
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import _capi
from .dynamics_and_models import DevArray, _dev, _resolve_device, _stream

__all__ = ['MLPNet', 'Policy4Toyota', 'LoadPolicy', 'orthogonal']


def orthogonal(rng, rows, cols, gain):
    """tf.keras.initializers.Orthogonal(gain) for a [rows, cols] kernel: QR of a normal matrix, sign-fixed."""
    a = rng.standard_normal((max(rows, cols), min(rows, cols)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if rows < cols:
        q = q.T
    return (gain * q[:rows, :cols]).astype(np.float32)


class MLPNet(object):
    """utils/model.py:18-43.  `hidden_activation` / `output_activation` in {'elu', 'relu', 'tanh', 'linear', None}."""

    def __init__(self, input_dim, num_hidden_layers, num_hidden_units, hidden_activation, output_dim, **kwargs):
        self.name = kwargs.get('name', 'mlp')
        self.input_dim, self.num_hidden_layers = int(input_dim), int(num_hidden_layers)
        self.num_hidden_units, self.output_dim = int(num_hidden_units), int(output_dim)
        self.hidden_activation = hidden_activation
        self.output_activation = kwargs.get('output_activation') or 'linear'
        for a in (self.hidden_activation, self.output_activation):
            if a not in _capi.ACT_ID:
                raise ValueError('unsupported activation %r' % (a,))
        dev = kwargs.get('device')
        self.device = _resolve_device(dev)
        self.api = _capi.hip_api()
        self._obs_scale = None
        self._handle = None
        rng = np.random.default_rng(kwargs.get('seed', 0))
        dims = [self.input_dim] + [self.num_hidden_units] * self.num_hidden_layers + [self.output_dim]
        w = []
        for L in range(self.num_hidden_layers + 1):      # Orthogonal(sqrt 2) hidden, Orthogonal(1) output, zero bias
            gain = np.sqrt(2.) if L < self.num_hidden_layers else 1.
            w += [orthogonal(rng, dims[L], dims[L + 1], gain), np.zeros((dims[L + 1],), np.float32)]
        self.set_weights(w)

    # -- weights ------------------------------------------------------------------------------
    def get_weights(self):
        return [a.copy() for a in self._weights]

    def set_weights(self, weights):
        weights = [np.ascontiguousarray(a, np.float32) for a in weights]
        if len(weights) != 2 * (self.num_hidden_layers + 1):
            raise ValueError('expected %d arrays (kernel, bias per Dense layer)' % (2 * (self.num_hidden_layers + 1)))
        self._weights = weights
        self._rebuild()

    def set_obs_scale(self, obs_scale):
        """Preprocessor 'scale' (utils/preprocessor.py:121): applied inside the kernel while staging the input."""
        self._obs_scale = None if obs_scale is None else np.ascontiguousarray(obs_scale, np.float32)
        self._rebuild()

    def _rebuild(self):
        layers = list(zip(self._weights[0::2], self._weights[1::2]))
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        new = self.api.mlp_create_from(self.input_dim, self.num_hidden_layers, self.num_hidden_units, self.output_dim,
                                       self.hidden_activation, self.output_activation, layers, self._obs_scale, index)
        if self._handle is not None:
            torch.cuda.synchronize(self.device)
            self.api.lib.eb_mlp_destroy(self._handle)
        self._handle = new

    def __del__(self):
        try:
            if self._handle is not None:
                self.api.lib.eb_mlp_destroy(self._handle)
        except Exception:
            pass

    # -- forward ------------------------------------------------------------------------------
    def _in(self, x):
        t = _dev(x, self.device)
        if t.dim() != 2 or t.shape[1] != self.input_dim:
            raise ValueError('input must be [B, %d], got %s' % (self.input_dim, tuple(t.shape)))
        return t

    def _stream(self):
        return _stream(self.device)

    def call(self, x, **kwargs):                       # utils/model.py:39-43
        t = self._in(x)
        out = torch.empty((t.shape[0], self.output_dim), dtype=torch.float32, device=self.device)
        self.api.mlp_forward(self._handle, t.shape[0], C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), self._stream())
        return DevArray(out)

    __call__ = call

    def mode(self, x, action_range):
        """action_range * tanh(mean) (utils/policy.py:68-72, 89-92) without materialising the logits."""
        t = self._in(x)
        out = torch.empty((t.shape[0], self.output_dim // 2), dtype=torch.float32, device=self.device)
        self.api.policy_run_batch(self._handle, t.shape[0], C.c_void_p(t.data_ptr()),
                                  C.c_float(-1.0 if action_range is None else float(action_range)),
                                  C.c_void_p(out.data_ptr()), self._stream())
        return DevArray(out)


class Policy4Toyota(object):
    """utils/policy.py:19-101, inference side.  `args` carries obs_dim, act_dim, num_hidden_layers, num_hidden_units,
    hidden_activation, policy_out_activation, action_range, deterministic_policy (as the experiment's config.json)."""

    def __init__(self, args, device=None, seed=0):
        self.args = args
        obs_dim, act_dim = int(args.obs_dim), int(args.act_dim)
        n_hiddens, n_units, act = int(args.num_hidden_layers), int(args.num_hidden_units), args.hidden_activation
        self.policy = MLPNet(obs_dim, n_hiddens, n_units, act, act_dim * 2, name='policy',
                             output_activation=getattr(args, 'policy_out_activation', 'linear'), device=device, seed=seed)
        self.obj_v = MLPNet(obs_dim, n_hiddens, n_units, act, 1, name='obj_v', output_activation='relu',
                            device=device, seed=seed + 1)
        self.models = (self.obj_v, self.policy,)

    def get_weights(self):
        return [model.get_weights() for model in self.models]

    def set_weights(self, weights):
        for i, weight in enumerate(weights):
            self.models[i].set_weights(weight)

    def save_weights(self, save_dir, iteration):
        os.makedirs(save_dir, exist_ok=True)
        arrays = {'%s_%d' % (m.name, k): a for m in self.models for k, a in enumerate(m.get_weights())}
        np.savez(os.path.join(save_dir, 'weights_ite%d.npz' % int(iteration)), **arrays)

    def load_weights(self, load_dir, iteration):
        z = np.load(os.path.join(load_dir, 'weights_ite%d.npz' % int(iteration)))
        for m in self.models:
            m.set_weights([z['%s_%d' % (m.name, k)] for k in range(2 * (m.num_hidden_layers + 1))])

    def compute_mode(self, obs):                       # utils/policy.py:68-72
        return self.policy.mode(obs, getattr(self.args, 'action_range', None))

    def compute_action(self, obs):                     # utils/policy.py:85-98
        if not getattr(self.args, 'deterministic_policy', True):
            raise NotImplementedError('sampling from the policy distribution is a training-side feature (out of scope)')
        return self.compute_mode(obs), 0.

    def compute_obj_v(self, obs):                      # utils/policy.py:100-103
        return DevArray(self.obj_v(obs).t[:, 0])


class LoadPolicy(object):
    """utils/load_policy.py:19-63.  `exp_dir/config.json` holds the arguments; the weights come from
    `exp_dir/models/weights_ite{iter}.npz` when present (else the random initialisation stands — useful for
    benchmarks).  The preprocessor is folded into the kernels, so run_batch / obj_value_batch take raw obs."""

    def __init__(self, exp_dir=None, iter=None, args=None, device=None):
        if args is None:
            params = json.loads(open(os.path.join(exp_dir, 'config.json')).read())
            args = SimpleNamespace(**params)
        elif isinstance(args, dict):
            args = SimpleNamespace(**args)
        self.args = args
        self.policy = Policy4Toyota(args, device=device)
        if exp_dir is not None and iter is not None:
            self.policy.load_weights(os.path.join(exp_dir, 'models'), iter)
        if getattr(args, 'obs_preprocess_type', 'scale') == 'scale' and getattr(args, 'obs_scale', None) is not None:
            scale = np.asarray(args.obs_scale, np.float32)
            for m in self.policy.models:
                m.set_obs_scale(scale)
        elif getattr(args, 'obs_preprocess_type', None) == 'normalize':
            raise NotImplementedError("obs_preprocess_type 'normalize' (running statistics) is not supported")

    def run_batch(self, obses):                        # utils/load_policy.py:53-57
        actions, _ = self.policy.compute_action(obses)
        return actions

    def obj_value_batch(self, obses):                  # utils/load_policy.py:59-63
        return self.policy.compute_obj_v(obses)

    __call__ = run_batch
