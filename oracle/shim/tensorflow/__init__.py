"""TEST INFRASTRUCTURE — container-only stand-in for the `tensorflow` module.

TensorFlow is not installed in this image, and the reference uses it
only as an fp32 elementwise/gather executor (27 symbols in dynamics_and_models.py).
This module supplies those symbols on top of NumPy so that /root/reference/*.py can be
imported UNMODIFIED by oracle/gen_golden.py to produce the golden vectors under
tests/golden/.  Every op rounds to fp32 exactly once, python scalars meeting a tensor
are cast to the tensor's dtype first (TF semantics), argmin returns the first minimum
(int64).  It never travels to the GPU box as part of any product path and nothing in
env_build_amd/ imports it.
"""
import contextlib

import numpy as np

float32 = np.float32
int32 = np.int32
int64 = np.int64


def _raw(x):
    return x.a if isinstance(x, Tensor) else x


def _coerce(other, like):
    """TF casts python scalars / lists to the tensor's dtype before the op."""
    other = _raw(other)
    if isinstance(other, np.ndarray):
        if other.dtype == np.float64 and like.dtype == np.float32:
            return other.astype(np.float32)
        return other
    if isinstance(other, (bool, np.bool_)):
        return other
    if isinstance(other, (int, float, np.floating, np.integer)):
        if like.dtype.kind == 'f':
            return np.asarray(other, dtype=like.dtype)
        if like.dtype.kind in 'iu' and isinstance(other, (int, np.integer)):
            return np.asarray(other, dtype=like.dtype)
        return np.asarray(other)
    return np.asarray(other)


class Tensor(object):
    __array_priority__ = 1000

    def __init__(self, a):
        a = _raw(a)
        a = np.asarray(a)
        if a.dtype == np.float64:
            a = a.astype(np.float32)
        self.a = a

    # -- plumbing
    def numpy(self):
        return self.a

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)

    def __len__(self):
        return len(self.a)

    def __getitem__(self, k):
        return Tensor(self.a[_raw(k)])

    def __iter__(self):
        for i in range(len(self.a)):
            yield Tensor(self.a[i])

    def __repr__(self):
        return 'shimTensor(%r)' % (self.a,)

    def __float__(self):
        return float(self.a)

    def __int__(self):
        return int(self.a)

    def __bool__(self):
        return bool(self.a)

    def __index__(self):
        return int(self.a)

    @property
    def shape(self):
        return self.a.shape

    @property
    def dtype(self):
        return self.a.dtype

    # -- arithmetic: one fp32 rounding per op
    def _bin(self, other, fn, rev=False):
        o = _coerce(other, self.a)
        r = fn(o, self.a) if rev else fn(self.a, o)
        return Tensor(r)

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.true_divide)
    def __rtruediv__(self, o): return self._bin(o, np.true_divide, True)
    def __neg__(self): return Tensor(-self.a)
    def __lt__(self, o): return self._bin(o, np.less)
    def __le__(self, o): return self._bin(o, np.less_equal)
    def __gt__(self, o): return self._bin(o, np.greater)
    def __ge__(self, o): return self._bin(o, np.greater_equal)
    def __eq__(self, o): return self._bin(o, np.equal)
    def __ne__(self, o): return self._bin(o, np.not_equal)
    __hash__ = None


def convert_to_tensor(x, dtype=None):
    a = np.asarray(_raw(x))
    if dtype is not None:
        a = a.astype(dtype)
    return Tensor(a)


def constant(x, dtype=None):
    if dtype is None:
        a = np.asarray(_raw(x))
        if a.dtype == np.int64:
            a = a.astype(np.int32)
        return Tensor(a)
    return convert_to_tensor(x, dtype)


def _f(x):
    a = np.asarray(_raw(x))
    if a.dtype == np.float64:
        a = a.astype(np.float32)
    return a


def square(x): return Tensor(np.square(_f(x)))
def sqrt(x): return Tensor(np.sqrt(_f(x)))
def sin(x): return Tensor(np.sin(_f(x)))
def cos(x): return Tensor(np.cos(_f(x)))
def atan(x): return Tensor(np.arctan(_f(x)))
def zeros_like(x): return Tensor(np.zeros_like(_f(x)))
def stop_gradient(x): return x if isinstance(x, Tensor) else Tensor(x)


def zeros(shape, dtype=np.float32):
    return Tensor(np.zeros(shape, dtype=dtype))


def cast(x, dtype):
    return Tensor(np.asarray(_raw(x)).astype(dtype))


def where(cond, x, y):
    c = np.asarray(_raw(cond))
    xa, ya = _raw(x), _raw(y)
    # python scalar branch adopts the dtype of the tensor branch (TF semantics)
    if not isinstance(xa, np.ndarray) and isinstance(ya, np.ndarray):
        xa = np.asarray(xa, dtype=ya.dtype)
    if not isinstance(ya, np.ndarray) and isinstance(xa, np.ndarray):
        ya = np.asarray(ya, dtype=xa.dtype)
    return Tensor(np.where(c, xa, ya))


def stack(xs, axis=0):
    return Tensor(np.stack([_f(x) for x in xs], axis))


def concat(xs, axis):
    return Tensor(np.concatenate([_f(x) for x in xs], axis))


def tile(x, multiples):
    return Tensor(np.tile(_raw(x), tuple(int(m) for m in np.asarray(_raw(multiples)))))


def reshape(x, shape):
    return Tensor(np.reshape(_raw(x), shape))


def shape(x):
    return Tensor(np.asarray(np.shape(_raw(x)), dtype=np.int32))


def expand_dims(x, axis):
    return Tensor(np.expand_dims(_raw(x), axis))


def gather(params, indices):
    return Tensor(np.asarray(_raw(params))[np.asarray(_raw(indices))])


def argmin(x, axis):
    return Tensor(np.argmin(_raw(x), axis).astype(np.int64))


def clip_by_value(x, lo, hi):
    a = _f(x)
    return Tensor(np.minimum(np.maximum(a, np.asarray(lo, a.dtype)), np.asarray(hi, a.dtype)))


def logical_and(a, b):
    return Tensor(np.logical_and(_raw(a), _raw(b)))


@contextlib.contextmanager
def name_scope(name):
    yield name


def function(fn=None, **kwargs):
    if fn is None:
        return lambda f: f
    return fn


class TensorSpec(object):
    def __init__(self, *a, **k):
        pass


class _Threading(object):
    @staticmethod
    def set_inter_op_parallelism_threads(n):
        pass

    @staticmethod
    def set_intra_op_parallelism_threads(n):
        pass


class _Experimental(object):
    @staticmethod
    def set_visible_devices(devices, kind=None):
        pass


class _Config(object):
    threading = _Threading()
    experimental = _Experimental()


config = _Config()


# ---- the inference surface of utils/model.py / utils/policy.py / utils/load_policy.py (tensorflow/keras holds the layers) ----
def tanh(x): return Tensor(np.tanh(_f(x)).astype(np.float32))
def exp(x): return Tensor(np.exp(_f(x)).astype(np.float32))


def split(x, num_or_size_splits, axis=0):
    return [Tensor(p) for p in np.split(_f(x), num_or_size_splits, axis=axis)]


def squeeze(x, axis=None):
    return Tensor(np.squeeze(_f(x), axis=axis))


class Module(object):
    def __init__(self, *a, **k):
        pass


class _Checkpoint(object):
    def __init__(self, **objects):
        self.objects = objects

    def save(self, path):
        raise RuntimeError('the stand-in has no checkpoint format')

    def restore(self, path):       # LoadPolicy.__init__ restores unconditionally; the fixtures set weights afterwards
        return self


class _Train(object):
    Checkpoint = _Checkpoint


train = _Train()
from . import keras  # noqa: E402  (tf.keras.* attribute access)
