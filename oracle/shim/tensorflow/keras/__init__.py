"""TEST INFRASTRUCTURE — the slice of tf.keras that the reference's utils/model.py and utils/policy.py touch (Model,
Sequential, Dense with string activations, two initializers, a do-nothing Adam), on NumPy fp32.  Dense is
activation(x @ kernel + bias) with float32 operands (utils/model.py:21-36 asks for dtype=tf.float32); weights are listed
kernel-then-bias per layer in construction order, as Keras' get_weights() does.  Container-only, see ../__init__.py."""
import types
import zlib

import numpy as np

from .. import Tensor, _raw


def _act(name):
    if name in (None, 'linear'):
        return lambda x: x
    if name == 'relu':
        return lambda x: np.maximum(x, np.float32(0))
    if name == 'tanh':
        return lambda x: np.tanh(x).astype(np.float32)
    if name == 'elu':
        return lambda x: np.where(x > 0, x, np.exp(np.minimum(x, np.float32(0))).astype(np.float32) - np.float32(1)).astype(np.float32)
    raise ValueError('activation %r is not part of the stand-in' % (name,))


class _Orthogonal(object):
    def __init__(self, gain=1.0, seed=None):
        self.gain = float(gain)

    def __call__(self, shape, rng):
        a = rng.standard_normal((max(shape), min(shape)))
        q, r = np.linalg.qr(a)
        q = q * np.sign(np.diag(r))
        q = q if shape[0] >= shape[1] else q.T
        return (self.gain * q[:shape[0], :shape[1]]).astype(np.float32)


class _Constant(object):
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, shape, rng):
        return np.full(shape, self.value, np.float32)


initializers = types.SimpleNamespace(Orthogonal=_Orthogonal, Constant=_Constant)


class Layer(object):
    def get_weights(self):
        return [w.copy() for w in self.weights]

    def set_weights(self, ws):
        ws = [np.asarray(w, np.float32) for w in ws]
        assert len(ws) == len(self.weights) and all(a.shape == b.shape for a, b in zip(ws, self.weights))
        for dst, src in zip(self.weights, ws):
            dst[...] = src

    def __call__(self, x, **kwargs):
        return self.call(x, **kwargs)


class Dense(Layer):
    def __init__(self, units, activation=None, kernel_initializer=None, bias_initializer=None, dtype=None, **kwargs):
        self.units, self.activation = int(units), _act(activation)
        self.kernel_initializer = kernel_initializer or _Orthogonal(1.0)
        self.bias_initializer = bias_initializer or _Constant(0.0)
        self.weights = []

    def build_for(self, in_dim, rng):
        self.weights = [self.kernel_initializer((in_dim, self.units), rng), self.bias_initializer((self.units,), rng)]
        return self.units

    def call(self, x, **kwargs):
        a = np.asarray(_raw(x), np.float32)
        return Tensor(self.activation((a @ self.weights[0] + self.weights[1]).astype(np.float32)))


class Sequential(Layer):
    def __init__(self, layers=None, **kwargs):
        self.layers = list(layers or [])

    @property
    def weights(self):
        return [w for l in self.layers for w in l.weights]

    def build_for(self, in_dim, rng):
        for l in self.layers:
            in_dim = l.build_for(in_dim, rng)
        return in_dim

    def call(self, x, **kwargs):
        for l in self.layers:
            x = l(x)
        return x


class Model(Layer):
    """attribute order = construction order = weight order, like a subclassed Keras model"""
    def __init__(self, name=None, **kwargs):
        object.__setattr__(self, '_sub', [])
        self.name = name

    def __setattr__(self, k, v):
        if isinstance(v, Layer) and k != '_sub':
            self._sub.append(v)
        object.__setattr__(self, k, v)

    @property
    def weights(self):
        return [w for l in self._sub for w in l.weights]

    def build(self, input_shape):
        rng = np.random.default_rng(zlib.crc32(str(self.name).encode()))   # deterministic per model name
        d = int(input_shape[-1])
        for l in self._sub:          # MLPNet chains its sub-layers in construction order (utils/model.py:39-43)
            d = l.build_for(d, rng)


from . import optimizers  # noqa: E402,F401  (tf.keras.optimizers.Adam)
