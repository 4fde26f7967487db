"""envbuild_binding.py — the reference-side binding of INTEGRATION.md §2: a ctypes stub over libenvbuild_hip.so that a
maintainer of idthanm/env_build drops next to dynamics_and_models.py and calls from EnvironmentModel's own methods.

It depends on nothing of env_build_amd/: only the shared library (include/envbuild.h is the contract), NumPy for the
host-side path tables and any framework that owns device memory (torch-ROCm here: `tensor.data_ptr()` and the current
stream).  tests/test_gpu_binding.py drives it against the CPU oracle, so what the document shows is what runs.

    ref_path  = ReferencePath(task)                       # the reference's own object (DAM:583): .path_list
    hip_model = HipEnvironmentModel(task, 0, 'training', ref_path, VEHICLE_MODE_LIST[task])
    obses, rewards, punish_train, punish_real, veh2veh4real, veh2road4real, actions = \
        hip_model.rollout_out(obses, raw_actions, ref_indexes, path_id)       # body of DAM:118-126
"""
import ctypes as C
import os

import numpy as np
import torch          # FIRST: PyTorch-ROCm bundles its HIP runtime; the library must bind to the same one

_P, _I = C.c_void_p, C.c_int32
_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.environ.get('ENVBUILD_LIB', os.path.join(_HERE, '..', 'env_build_amd', 'lib', 'libenvbuild_hip.so'))
VMODE = {m: i for i, m in enumerate(('dl', 'du', 'dr', 'rd', 'rl', 'ru', 'ur', 'ud', 'ul', 'lu', 'lr', 'ld'))}   # EB_VMODE_*
TASK = {'left': 0, 'straight': 1, 'right': 2}                                                                 # EB_TASK_*
BINDING_ABI = 5     # the EB_ABI_VERSION of the include/envbuild.h these prototypes were written against


class _Cfg(C.Structure):       # struct eb_config
    _fields_ = [(n, C.c_int32) for n in ('abi_version', 'task', 'n_veh', 'n_future', 'mode', 'device')]


def load(path=DEFAULT_LIB):
    """CDLL with argtypes on every entry point used below: without them ctypes passes Python ints as C `int` and a
    64-bit address is truncated."""
    lib = C.CDLL(os.path.abspath(path))
    lib.eb_last_error.restype = C.c_char_p
    lib.eb_abi_version.restype = C.c_int
    lib.eb_create.argtypes = [C.POINTER(_Cfg), C.POINTER(_P)]
    lib.eb_destroy.argtypes = [_P]
    lib.eb_set_paths.argtypes = [_P, _P, _P, _P, _P, _I]
    lib.eb_set_veh_modes.argtypes = [_P, _P, _I]
    lib.eb_rollout_step.argtypes = [_P, _I, _P, _P, _P, _I, _P, _P, _P, _P]
    lib.eb_compute_rewards.argtypes = [_P, _I, _P, _P, _P, _P, _P]
    for fn in (lib.eb_create, lib.eb_destroy, lib.eb_set_paths, lib.eb_set_veh_modes, lib.eb_rollout_step,
               lib.eb_compute_rewards):
        fn.restype = C.c_int
    return lib


def _host(a):
    return a.ctypes.data_as(_P)


def _dev(t):
    return None if t is None else _P(t.data_ptr())


class HipEnvironmentModel(object):
    """What EnvironmentModel.__init__ (DAM:91-106) would create once per (task, device)."""

    def __init__(self, task, num_future_data, mode, ref_path, veh_mode_list, device=0, lib=None):
        self.lib = lib if lib is not None else load()
        if self.lib.eb_abi_version() != BINDING_ABI:     # a library with other prototypes: refuse, do not call it with shifted arguments
            raise RuntimeError('libenvbuild: ABI %d, this binding was written for %d' % (self.lib.eb_abi_version(), BINDING_ABI))
        cfg = _Cfg(BINDING_ABI, TASK[task], len(veh_mode_list), num_future_data, 0 if mode == 'training' else 1, device)
        self.h = _P()
        self._ck(self.lib.eb_create(C.byref(cfg), C.byref(self.h)))
        xs, ys, ph = (np.ascontiguousarray(np.concatenate([np.asarray(p[k], np.float32) for p in ref_path.path_list]))
                      for k in range(3))                                         # the tables of DAM:598-700
        lens = np.array([len(p[0]) for p in ref_path.path_list], np.int32)
        self._ck(self.lib.eb_set_paths(self.h, _host(xs), _host(ys), _host(ph), _host(lens), len(lens)))
        ids = np.array([VMODE[m] for m in veh_mode_list], np.uint8)              # VEHICLE_MODE_LIST[task], UTL:44-46
        self._ck(self.lib.eb_set_veh_modes(self.h, _host(ids), len(ids)))
        self.D = 6 + 3 * (num_future_data + 1) + 4 * len(veh_mode_list)

    def _ck(self, rc):
        if rc:
            raise RuntimeError((self.lib.eb_last_error() or b'').decode())

    def rollout_out(self, obses, actions, ref_indexes, path_id=0):
        """Body of EnvironmentModel.rollout_out (DAM:118-126): obses [B, D], actions [B, 2] raw, fp32, on the GPU."""
        B = obses.shape[0]
        out, out5 = torch.empty_like(obses), torch.empty((5, B), dtype=torch.float32, device=obses.device)
        scaled = torch.empty_like(actions)
        stream = _P(torch.cuda.current_stream().cuda_stream)
        self._ck(self.lib.eb_rollout_step(self.h, B, _dev(obses), _dev(actions), _dev(ref_indexes), int(path_id),
                                          _dev(out), _dev(out5), _dev(scaled), stream))
        return out, out5[0], out5[1], out5[2], out5[3], out5[4], scaled     # self.obses, rewards, ..., self.actions

    def compute_rewards(self, obses, scaled_actions):
        """Body of EnvironmentModel.compute_rewards (DAM:186-320): the 5 outputs + the 16 reward_dict terms [16, B]."""
        B = obses.shape[0]
        out5 = torch.empty((5, B), dtype=torch.float32, device=obses.device)
        d16 = torch.empty((16, B), dtype=torch.float32, device=obses.device)
        stream = _P(torch.cuda.current_stream().cuda_stream)
        self._ck(self.lib.eb_compute_rewards(self.h, B, _dev(obses), _dev(scaled_actions), _dev(out5), _dev(d16), stream))
        return out5, d16

    def close(self):
        if self.h:
            self.lib.eb_destroy(self.h)
            self.h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass
