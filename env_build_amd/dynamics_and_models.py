"""Drop-in counterparts of the reference's dynamics_and_models.py classes — VehicleDynamics,
EnvironmentModel, ReferencePath — with the arithmetic running in HIP kernels on an MI355X.

Same class / method / attribute names and argument meaning as the reference (file:line cited per
method; DAM = /root/reference/dynamics_and_models.py).  Inputs may be NumPy arrays, torch tensors
(CPU or ROCm) or the DevArray values these classes return; outputs are DevArray — a thin view of a
device tensor that offers what the reference's callers use on tf.Tensor: `.numpy()`, indexing,
arithmetic, len().  torch is used only for device memory and streams.

There is no CPU path: constructing any of these without libenvbuild_hip.so and a GPU raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from .endtoend_env_utils import EXPECTED_V, VEHICLE_MODE_LIST, tiled_mode_list
from .ref_path_tables import build_ref_paths

__all__ = ['VehicleDynamics', 'EnvironmentModel', 'ReferencePath', 'DevArray', 'deal_with_phi_diff']


# ----------------------------------------------------------------------------------------------
# value type
# ----------------------------------------------------------------------------------------------
class DevArray(object):
    """A device tensor with the slice of the tf.Tensor surface the reference's callers rely on
    (E2E:280, 297, 506-507; hier_decision.py:96, 107)."""
    __array_priority__ = 1000
    __slots__ = ('t',)

    def __init__(self, t):
        self.t = t

    def numpy(self):
        return self.t.detach().cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def torch(self):
        return self.t

    def __len__(self):
        return len(self.t)

    def __getitem__(self, k):
        return DevArray(self.t[_unwrap(k)])

    def __iter__(self):
        for i in range(len(self.t)):
            yield DevArray(self.t[i])

    def __repr__(self):
        return 'DevArray(%r)' % (self.t,)

    def __float__(self):
        return float(self.t)

    def __int__(self):
        return int(self.t)

    def __bool__(self):
        return bool(self.t)

    @property
    def shape(self):
        return tuple(self.t.shape)

    @property
    def dtype(self):
        return self.t.dtype

    @property
    def device(self):
        return self.t.device

    def data_ptr(self):
        return self.t.data_ptr()

    def _b(self, o, fn, rev=False):
        o = _unwrap(o)
        if isinstance(o, np.ndarray):
            o = torch.from_numpy(o).to(self.t.device)
        return DevArray(fn(o, self.t) if rev else fn(self.t, o))

    def __add__(self, o): return self._b(o, torch.add)
    def __radd__(self, o): return self._b(o, torch.add, True)
    def __sub__(self, o): return self._b(o, torch.sub)
    def __rsub__(self, o): return self._b(o, torch.sub, True)
    def __mul__(self, o): return self._b(o, torch.mul)
    def __rmul__(self, o): return self._b(o, torch.mul, True)
    def __truediv__(self, o): return self._b(o, torch.div)
    def __rtruediv__(self, o): return self._b(o, torch.div, True)
    def __neg__(self): return DevArray(-self.t)
    def __lt__(self, o): return self._b(o, torch.lt)
    def __le__(self, o): return self._b(o, torch.le)
    def __gt__(self, o): return self._b(o, torch.gt)
    def __ge__(self, o): return self._b(o, torch.ge)
    def __eq__(self, o): return self._b(o, torch.eq)
    def __ne__(self, o): return self._b(o, torch.ne)
    __hash__ = None


class _RowOf(DevArray):
    """Row k of a [rows, B] device tensor (one of rollout_out's five per-env outputs) — or, with shape given, a reshaped window of it —
    as a DevArray whose view tensor is made when the value is first used: handing out five of them costs five small Python
    objects per call, not five tensor-indexing operations."""

    def __init__(self, base, k, rows=1, shape=None):
        self._base, self._k, self._rows, self._shape = base, k, rows, shape
        self._t = None

    @property
    def t(self):
        if self._t is None:
            b = self._base
            self._t = b[self._k] if self._shape is None else b[self._k:self._k + self._rows].view(self._shape)
        return self._t

    @t.setter
    def t(self, v):
        self._t = v


class _OutSet(object):
    """One set of rollout_out's outputs: the next obs [B, D], and ONE [7, B] tensor holding the five per-env outputs (rows 0-4,
    the C entry's out5) and the scaled actions (rows 5-6 viewed as [B, 2]); the device addresses the C call needs; the 6-tuple."""
    __slots__ = ('obs', 'out7', 'obs_ptr', 'out5_ptr', 'scaled_ptr', 'ret', 'actions', 'obses')

    def __init__(self, like, B, device):
        self.obs = torch.empty_like(like)
        self.out7 = torch.empty((7, B), dtype=torch.float32, device=device)
        self.obs_ptr = self.obs.data_ptr()
        self.out5_ptr = self.out7.data_ptr()
        self.scaled_ptr = self.out5_ptr + 5 * B * 4
        self.obses = DevArray(self.obs)
        o7 = self.out7
        self.ret = (self.obses, _RowOf(o7, 0), _RowOf(o7, 1), _RowOf(o7, 2), _RowOf(o7, 3), _RowOf(o7, 4))
        self.actions = _RowOf(o7, 5, 2, (B, 2))


def _unwrap(x):
    return x.t if isinstance(x, DevArray) else x


def _default_device():
    if not torch.cuda.is_available():
        raise _capi.EbError('env_build_amd needs an MI355X visible to PyTorch-ROCm (torch.cuda.is_available() is '
                            'False); there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _resolve_device(device):
    """torch.device with an explicit index.  A handle (tables, scratch) lives on ONE device: a bare 'cuda' is the device current NOW, for
    the object's whole life — a later torch.cuda.set_device(other) changes neither where it launches nor whose current stream orders
    its launches."""
    dev = torch.device(device) if device is not None else _default_device()
    if dev.type == 'cuda' and dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    return dev


def _dev(x, device, dtype=torch.float32):
    """numpy / torch / DevArray -> contiguous tensor of `dtype` on `device`."""
    x = _unwrap(x)
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous()
    np_dtype = {torch.float32: np.float32, torch.int32: np.int32, torch.uint8: np.uint8}[dtype]
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np_dtype)).to(device)


def _ptr(t):
    """device address for a ctypes call (every entry point has argtypes, so a plain int converts to void*: no c_void_p object per
    argument — 27 of them per step + reset)"""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream(device):
    """the hipStream_t PyTorch currently launches on for `device` (what `with torch.cuda.stream(s):` sets), as an int"""
    if _raw_stream is not None:
        return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())   # ~0.2 us; current_stream(): ~5
    return torch.cuda.current_stream(device).cuda_stream


class _Handle(object):
    """Owns one eb_handle (destroyed with the object)."""

    def __init__(self, task, n_veh, n_future, mode, device, modes=None, with_paths=True):
        self.api = _capi.hip_api()
        self.device = device
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self.h = self.api.create(task, n_veh, n_future, mode, idx)
        if with_paths:
            paths = _tables(task)[0]
            xs = np.ascontiguousarray(np.concatenate([p[0] for p in paths]), np.float32)
            ys = np.ascontiguousarray(np.concatenate([p[1] for p in paths]), np.float32)
            ph = np.ascontiguousarray(np.concatenate([p[2] for p in paths]), np.float32)
            lens = np.array([len(p[0]) for p in paths], np.int32)
            self.api.set_paths(self.h, xs.ctypes.data_as(C.c_void_p), ys.ctypes.data_as(C.c_void_p),
                               ph.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), len(paths))
        if modes is not None:
            ids = np.array([_capi.VMODE_ID[m] for m in modes], np.uint8)
            self.api.set_veh_modes(self.h, ids.ctypes.data_as(C.c_void_p), len(ids))

    def __del__(self):
        try:
            self.api.destroy(self.h)
        except Exception:
            pass


_TABLES = {}
_UTIL = {}


def _tables(task):
    if task not in _TABLES:
        _TABLES[task] = build_ref_paths(task)
    return _TABLES[task]


def _util_handle(device, task='left', mode_name=None):
    """Cached single-slot handles for the ops that need no model state (f_xu, predict_for_a_mode,
    ReferencePath queries)."""
    key = (str(device), task, mode_name)
    if key not in _UTIL:
        _UTIL[key] = _Handle(task, 1, 0, _capi.MODE_SELECTING, device, modes=[mode_name or 'du'])
    return _UTIL[key]


def deal_with_phi_diff(phi_diff):  # DAM:577-580
    t = _unwrap(phi_diff)
    dev = t.device if isinstance(t, torch.Tensor) and t.is_cuda else _default_device()
    x = _dev(t, dev)
    out = torch.empty_like(x)
    hd = _util_handle(dev)
    hd.api.phi_diff(hd.h, x.numel(), _ptr(x), _ptr(out), _stream(dev))
    return DevArray(out)


# ----------------------------------------------------------------------------------------------
class VehicleDynamics(object):
    """DAM:26-87."""

    def __init__(self, device=None):
        self.vehicle_params = dict(C_f=-155495.0, C_r=-155495.0, a=1.19, b=1.46, mass=1520., I_z=2642.,
                                   miu=0.8, g=9.81)                                   # DAM:37-45
        a, b, mass, g = (self.vehicle_params[k] for k in ('a', 'b', 'mass', 'g'))
        F_zf, F_zr = b * mass * g / (a + b), a * mass * g / (a + b)                   # DAM:48 (float64)
        self.vehicle_params.update(dict(F_zf=F_zf, F_zr=F_zr))
        self._device = device

    def _dev(self):
        return self._device if self._device is not None else _default_device()

    def f_xu(self, states, actions, tau):  # DAM:52-83
        dev = self._dev()
        hd = _util_handle(dev)
        st, ac = _dev(states, dev), _dev(actions, dev)
        n = st.shape[0]
        nxt = torch.empty((n, 6), dtype=torch.float32, device=dev)
        par = torch.empty((n, 4), dtype=torch.float32, device=dev)
        hd.api.f_xu(hd.h, n, _ptr(st), _ptr(ac), float(tau), _ptr(nxt), _ptr(par), _stream(dev))
        return DevArray(nxt), DevArray(par)

    def prediction(self, x_1, u_1, frequency):  # DAM:85-87
        return self.f_xu(x_1, u_1, 1 / frequency)


# ----------------------------------------------------------------------------------------------
class ReferencePath(object):
    """DAM:583-770.  The three candidate paths are built on the host exactly as the reference does
    (ref_path_tables.py) and uploaded once per (task, device)."""

    def __init__(self, task, ref_index=None, device=None):
        self.exp_v = EXPECTED_V
        self.task = task
        self.path_list, self.path_len_list, self.control_points = _tables(task)
        self.path_list = list(self.path_list)
        self.ref_index = np.random.choice(len(self.path_list)) if ref_index is None else ref_index   # DAM:591
        self.path = self.path_list[self.ref_index]
        self._device = device
        self._dev_tables = None

    def set_path(self, path_index=None):  # DAM:594-596
        self.ref_index = path_index
        self.path = self.path_list[self.ref_index]

    # -- helpers
    def _dev(self):
        return self._device if self._device is not None else _default_device()

    def _hd(self):
        return _util_handle(self._dev(), self.task)

    def _current_path_id(self):
        for k, p in enumerate(self.path_list):
            if p is self.path:
                return k
        return None

    def _query(self):
        """(handle, path id) the current `path` is answered from: the task's shared tables, or — when `path` was
        assigned a custom (xs, ys, phis) triple — a private one-path handle built on first use (the reference accepts
        any path there, DAM:594-596, E2E:793-795)."""
        k = self._current_path_id()
        if k is not None:
            return self._hd(), k
        cached = getattr(self, '_custom', None)
        if cached is None or cached[0] is not self.path:
            xs, ys, ph = (np.ascontiguousarray(np.asarray(_unwrap(c) if not isinstance(_unwrap(c), torch.Tensor)
                                                          else _unwrap(c).cpu().numpy()), np.float32) for c in self.path)
            if not (xs.ndim == ys.ndim == ph.ndim == 1 and len(xs) == len(ys) == len(ph)):
                raise ValueError('ReferencePath.path must be three 1-D arrays of equal length')
            hd = _Handle(self.task, 1, 0, _capi.MODE_SELECTING, self._dev(), modes=['du'], with_paths=False)
            lens = np.array([len(xs)], np.int32)
            hd.api.set_paths(hd.h, xs.ctypes.data_as(C.c_void_p), ys.ctypes.data_as(C.c_void_p),
                             ph.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), 1)
            self._custom = (self.path, hd)
        return self._custom[1], 0

    def find_closest_point(self, xs, ys, ratio=10):  # DAM:702-715
        dev = self._dev()
        hd, pid = self._query()
        x, y = _dev(xs, dev), _dev(ys, dev)
        n = x.shape[0]
        idx = torch.empty((n,), dtype=torch.int32, device=dev)
        pts = torch.empty((3, n), dtype=torch.float32, device=dev)
        hd.api.find_closest_point(hd.h, n, _ptr(x), _ptr(y), None, pid, int(ratio), _ptr(idx), _ptr(pts), _stream(dev))
        return DevArray(idx.to(torch.int64)), (DevArray(pts[0]), DevArray(pts[1]), DevArray(pts[2]))

    def _points(self, indexs, n_future):
        dev = self._dev()
        hd, pid = self._query()
        i = _unwrap(indexs)
        if not isinstance(i, torch.Tensor):
            i = torch.from_numpy(np.ascontiguousarray(np.asarray(i).reshape(-1)))
        i = i.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
        out = torch.empty((n_future + 1, 3, i.shape[0]), dtype=torch.float32, device=dev)
        hd.api.path_points(hd.h, i.shape[0], _ptr(i), None, pid, int(n_future), _ptr(out), _stream(dev))
        return out

    def indexs2points(self, indexs):  # DAM:726-733 (eb_path_points: a clamped gather)
        o = self._points(indexs, 0)[0]
        return DevArray(o[0]), DevArray(o[1]), DevArray(o[2])

    def future_n_data(self, current_indexs, n):  # DAM:717-724 (eb_path_points: + 80 per point, clamped to len - 2)
        o = self._points(current_indexs, n)
        return [(DevArray(o[k][0]), DevArray(o[k][1]), DevArray(o[k][2])) for k in range(1, n + 1)]

    def tracking_error_vector(self, ego_xs, ego_ys, ego_phis, ego_vs, n):  # DAM:735-770
        return self.tracking_error_vector_batched(ego_xs, ego_ys, ego_phis, ego_vs, n, ref_indexes=None)

    def tracking_error_vector_batched(self, ego_xs, ego_ys, ego_phis, ego_vs, n, ref_indexes=None):
        """tracking_error_vector with an optional per-row path id (what compute_next_obses'
        training-mode loop over path_list + tf.where computes, DAM:342-353)."""
        dev = self._dev()
        hd, pid = (self._hd(), 0) if ref_indexes is not None else self._query()
        x, y, ph, v = _dev(ego_xs, dev), _dev(ego_ys, dev), _dev(ego_phis, dev), _dev(ego_vs, dev)
        ri = None if ref_indexes is None else _dev(ref_indexes, dev, torch.int32)
        rows = x.shape[0]
        out = torch.empty((rows, 3 * (n + 1)), dtype=torch.float32, device=dev)
        hd.api.tracking_error(hd.h, rows, _ptr(x), _ptr(y), _ptr(ph), _ptr(v), _ptr(ri), pid, int(n), _ptr(out),
                              _stream(dev))
        return DevArray(out)


# ----------------------------------------------------------------------------------------------
class EnvironmentModel(object):  # DAM:90-427
    """Batched analytic model of the crossroad env.  Beyond the reference's signature:
    `n_veh` (vehicle slots per env; default = the task's native 8/9/5, other counts tile
    VEHICLE_MODE_LIST[task]) and `device`."""

    def __init__(self, training_task, num_future_data=0, mode='training', n_veh=None, device=None,
                 state_dtype='float32', copy_outputs=True):
        """copy_outputs: True (default) — what rollout_out hands out are arrays of their own, as in the reference: they can be kept
        in a list.  False — the outputs live in two pre-allocated sets used in turn (zero allocations per call; the reference's
        callers consume them at once, hier_decision.py:91-96): a value handed out by a call stays valid until the call AFTER the
        next one and must be consumed or copied by then."""
        if training_task not in ('left', 'straight', 'right'):
            raise ValueError("training_task must be 'left', 'straight' or 'right'")
        if state_dtype not in ('float32', 'float16'):
            raise ValueError("state_dtype must be 'float32' or 'float16'")
        # 'float16': self.obses is STORED as binary16 (BASELINE configs[4]); rollout_out widens each row to fp32,
        # runs the same fp32 arithmetic, returns fp32 rewards / penalties and rounds the next obs to binary16
        self.state_dtype = torch.float16 if state_dtype == 'float16' else torch.float32
        self.task = training_task
        self.mode = mode
        self.device = _resolve_device(device)
        self.vehicle_dynamics = VehicleDynamics(self.device)
        self.base_frequency = 10.
        self.obses = None
        self.ego_params = None
        self.actions = None
        self.ref_path = ReferencePath(self.task, device=self.device)
        self.ref_indexes = None
        self.num_future_data = num_future_data
        self.exp_v = EXPECTED_V
        self.reward_info = None
        self.ego_info_dim = 6
        self.per_veh_info_dim = 4
        self.per_tracking_info_dim = 3
        native = VEHICLE_MODE_LIST[self.task]
        self.veh_num = len(native) if n_veh is None else int(n_veh)
        self.veh_mode_list = list(native) if self.veh_num == len(native) else tiled_mode_list(self.task, self.veh_num)
        self.obs_dim = 6 + 3 * (num_future_data + 1) + 4 * self.veh_num
        self._hd = _Handle(self.task, self.veh_num, num_future_data,
                           _capi.MODE_TRAINING if mode == 'training' else _capi.MODE_SELECTING, self.device,
                           modes=self.veh_mode_list)
        self.api, self.handle = self._hd.api, self._hd.h
        self._ref_idx_dev = None
        # the raw entry point of the hot call (the checked wrapper costs a Python closure per call)
        self._step_fn = (self.api.lib.eb_rollout_step_f16 if self.state_dtype == torch.float16
                         else self.api.lib.eb_rollout_step)
        self.copy_outputs = bool(copy_outputs)
        self._sets, self._set_i = None, 0
        self._dev_index = self.device.index          # (explicit since the normalisation above: the handle's device)

    # -- state ------------------------------------------------------------------------------
    def _obs(self, obses, dtype=torch.float32):
        t = _unwrap(obses)
        if dtype == torch.float16:
            t = (t if isinstance(t, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(t))))
            t = t.to(device=self.device, dtype=torch.float16).contiguous()
        else:
            t = _dev(t, self.device)
        if t.dim() != 2 or t.shape[1] != self.obs_dim:
            raise ValueError('obses must be [B, %d] for task=%s, n_veh=%d, num_future_data=%d; got %s'
                             % (self.obs_dim, self.task, self.veh_num, self.num_future_data, tuple(t.shape)))
        return t

    def reset(self, obses, ref_indexes=None):  # DAM:108-112
        self.obses = DevArray(self._obs(obses, self.state_dtype))
        self.ref_indexes = ref_indexes
        self._ref_idx_dev = None if ref_indexes is None else _dev(ref_indexes, self.device, torch.int32)
        self.actions = None
        self.reward_info = None

    def add_traj(self, obses, path_index):  # DAM:114-116
        self.obses = DevArray(self._obs(obses, self.state_dtype))
        self.ref_path.set_path(path_index)

    def _path_args(self):
        if self.mode == 'training':
            if self._ref_idx_dev is None:
                raise ValueError("mode='training' needs ref_indexes: call reset(obses, ref_indexes) (DAM:344)")
            return self._ref_idx_dev, 0
        k = self.ref_path._current_path_id()
        if k is None:
            raise ValueError('EnvironmentModel tracks the paths of its task (ref_path.path_list); set one with '
                             'ref_path.set_path(i) / add_traj(obses, i)')
        return None, k

    def _after_tracking(self):
        if self.mode == 'training':   # the reference's loop leaves ref_path.path on the last path, DAM:345-346
            self.ref_path.path = self.ref_path.path_list[-1]

    # -- the hot path -----------------------------------------------------------------------
    def rollout_out(self, actions):  # DAM:118-126
        """One C call, one kernel launch.  Per call on the host: the argument checks, (copy_outputs) two allocations, one ctypes
        call; the 6-tuple's members are DevArrays whose row views are made on first use."""
        obs = self.obses.t if type(self.obses) is DevArray else None
        if obs is None or obs.dtype != self.state_dtype or obs.device != self.device or not obs.is_contiguous():
            obs = self._obs(self.obses, self.state_dtype)
        act = _unwrap(actions)
        if not (isinstance(act, torch.Tensor) and act.dtype == torch.float32 and act.device == self.device
                and act.is_contiguous()):
            act = _dev(act, self.device)
        B = obs.shape[0]
        if self.mode == 'training':
            ri = self._ref_idx_dev
            if ri is None:
                raise ValueError("mode='training' needs ref_indexes: call reset(obses, ref_indexes) (DAM:344)")
            ri_ptr, pid = ri.data_ptr(), 0
        else:
            ri, pid = self._path_args()
            ri_ptr = None
        if self.copy_outputs:
            st = _OutSet(obs, B, self.device)
        else:
            sets = self._sets
            if sets is None or sets[0].obs.shape != obs.shape or sets[0].obs.dtype != obs.dtype:
                sets = self._sets = [_OutSet(obs, B, self.device) for _ in range(2)]
            self._set_i ^= 1
            st = sets[self._set_i]
            if st.obs_ptr == obs.data_ptr():            # (the caller put one of our own arrays back as the state: the other set)
                self._set_i ^= 1
                st = sets[self._set_i]
        rc = self._step_fn(self.handle, B, obs.data_ptr(), act.data_ptr(), ri_ptr, pid, st.obs_ptr, st.out5_ptr, st.scaled_ptr,
                           _raw_stream(self._dev_index) if _raw_stream is not None else _stream(self.device))
        if rc != 0:
            self.api.check(rc)
        self.actions = st.actions
        self.obses = st.obses
        if self.mode == 'training':   # the reference's loop leaves ref_path.path on the last path, DAM:345-346
            self.ref_path.path = self.ref_path.path_list[-1]
        return st.ret

    def rollout_tape(self, action_tape):
        """Open-loop rollout over an action tape [H, B, 2] (the MPC callers' cost_function,
        mpc/main.py:470-479): H launches enqueued by one C call.  Returns (final obses,
        out5 [H, 5, B])."""
        obs = self._obs(self.obses, self.state_dtype)
        tape = _dev(action_tape, self.device)
        H, B = tape.shape[0], obs.shape[0]
        ri, pid = self._path_args()
        work, out = torch.empty_like(obs), torch.empty_like(obs)
        out5 = torch.empty((H, 5, B), dtype=torch.float32, device=self.device)
        tape_fn = self.api.rollout_tape_f16 if self.state_dtype == torch.float16 else self.api.rollout_tape
        tape_fn(self.handle, B, H, _ptr(obs), _ptr(tape), _ptr(ri), pid, _ptr(work), _ptr(out), _ptr(out5),
                _stream(self.device))
        self.obses = DevArray(out)
        self._after_tracking()
        return self.obses, DevArray(out5)

    def _action_transformation_for_end2end(self, actions):  # DAM:128-132
        act = _dev(actions, self.device)
        out = torch.empty_like(act)
        self.api.action_transform(self.handle, act.shape[0], _ptr(act), _ptr(out), _stream(self.device))
        return DevArray(out)

    def ss(self, obses, actions, lam=0.1):  # DAM:134-184
        obs, act = self._obs(obses), _dev(actions, self.device)
        ri, pid = self._path_args()
        out = torch.empty((obs.shape[0],), dtype=torch.float32, device=self.device)
        self.api.ss(self.handle, obs.shape[0], _ptr(obs), _ptr(act), _ptr(ri), pid, float(lam), _ptr(out),
                    _stream(self.device))
        self._after_tracking()
        return DevArray(out)

    REWARD_KEYS = ('punish_steer', 'punish_a_x', 'punish_yaw_rate', 'devi_v', 'devi_y', 'devi_phi',
                   'scaled_punish_steer', 'scaled_punish_a_x', 'scaled_punish_yaw_rate', 'scaled_devi_v',
                   'scaled_devi_y', 'scaled_devi_phi', 'veh2veh4training', 'veh2road4training', 'veh2veh4real',
                   'veh2road4real')  # DAM:302-318

    def compute_rewards(self, obses, actions):  # DAM:186-320
        obs, act = self._obs(obses), _dev(actions, self.device)
        B = obs.shape[0]
        out5 = torch.empty((5, B), dtype=torch.float32, device=self.device)
        d16 = torch.empty((16, B), dtype=torch.float32, device=self.device)
        self.api.compute_rewards(self.handle, B, _ptr(obs), _ptr(act), _ptr(out5), _ptr(d16), _stream(self.device))
        reward_dict = {k: DevArray(d16[i]) for i, k in enumerate(self.REWARD_KEYS)}
        return (DevArray(out5[0]), DevArray(out5[1]), DevArray(out5[2]), DevArray(out5[3]), DevArray(out5[4]),
                reward_dict)

    def compute_next_obses(self, obses, actions):  # DAM:322-358
        obs, act = self._obs(obses), _dev(actions, self.device)
        ri, pid = self._path_args()
        out = torch.empty_like(obs)
        self.api.compute_next_obses(self.handle, obs.shape[0], _ptr(obs), _ptr(act), _ptr(ri), pid, _ptr(out),
                                    _stream(self.device))
        self._after_tracking()
        return DevArray(out)

    def ego_predict(self, ego_infos, actions):  # DAM:386-392
        ego = _dev(ego_infos, self.device)[:, :6].contiguous()
        act = _dev(actions, self.device)
        out = torch.empty_like(ego)
        self.api.ego_predict(self.handle, ego.shape[0], _ptr(ego), _ptr(act), _ptr(out), _stream(self.device))   # f_xu + clip, DAM:387-390
        return DevArray(out)

    def veh_predict(self, veh_infos):  # DAM:394-403
        veh = _dev(veh_infos, self.device)
        if veh.dim() != 2 or veh.shape[1] != 4 * self.veh_num:
            raise ValueError('veh_infos must be [B, %d]' % (4 * self.veh_num))
        out = torch.empty_like(veh)
        self.api.veh_predict(self.handle, veh.shape[0], _ptr(veh), _ptr(out), _stream(self.device))
        return DevArray(out)

    def predict_for_a_mode(self, vehs, mode):  # DAM:405-427
        if mode not in _capi.VMODE_ID:
            raise ValueError('unknown vehicle mode %r' % (mode,))
        hd = _util_handle(self.device, 'left', mode)
        veh = _dev(vehs, self.device)
        out = torch.empty_like(veh)
        hd.api.veh_predict(hd.h, veh.shape[0], _ptr(veh), _ptr(out), _stream(self.device))
        return DevArray(out)

    def render(self, mode='human'):  # DAM:429-574 is a matplotlib debug view: out of scope
        raise NotImplementedError('EnvironmentModel.render (matplotlib debug view) is out of scope')
