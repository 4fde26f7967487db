"""ctypes binding of include/envbuild.h.

`CApi(path)` binds one shared library exporting the C-ABI.  The product only ever binds
env_build_amd/lib/libenvbuild_hip.so through `hip_api()`, which raises if the library is missing
or fails to load — there is no CPU fallback anywhere in the package.  (tests/ bind the CPU oracle
with the same class to compare the two libraries call-for-call.)
"""
import ctypes as C
import os

EB_ABI_VERSION = 5
TASK_ID = {'left': 0, 'straight': 1, 'right': 2}
MODE_TRAINING, MODE_SELECTING = 0, 1
# vehicle mode ids (EB_VMODE_*), in the order of the twelve lists of E2E:354
VMODES = ('dl', 'du', 'dr', 'rd', 'rl', 'ru', 'ur', 'ud', 'ul', 'lu', 'lr', 'ld')
VMODE_ID = {m: i for i, m in enumerate(VMODES)}
VMODE_EMPTY = 255
EXITS = ('D', 'R', 'U', 'L')                    # EB_EXIT_*: multi_ego.py:33 ROTATE_ANGLE = 0 / 90 / 180 / -90 degrees
EXIT_ID = {e: i for i, e in enumerate(EXITS)}

DONE_NAMES = ('not_done_yet', 'collision', 'break_road_constrain', 'deviate_too_much',
              'break_stability', 'break_red_light', 'good_done',  # E2E:208-221
              'time_limit')      # EB_DONE_TIME_LIMIT: gym's TimeLimit (README.md:55-59), not a reference outcome


class EbConfig(C.Structure):
    _fields_ = [('abi_version', C.c_int32), ('task', C.c_int32), ('n_veh', C.c_int32),
                ('n_future', C.c_int32), ('mode', C.c_int32), ('device', C.c_int32)]


class EbMlpConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('abi_version', 'obs_dim', 'n_hidden', 'n_units', 'out_dim',
                                         'hidden_act', 'out_act', 'device')]


class EbRespawn(C.Structure):     # struct eb_respawn: the pool's re-entry rule as the last stage of eb_env_step
    _fields_ = [('entry', C.c_void_p), ('limit', C.c_float), ('span', C.c_float), ('v_max', C.c_float),
                ('seed', C.c_uint64), ('counter', C.c_uint64), ('edge_span', C.c_float)]


class EbAutoReset(C.Structure):   # struct eb_auto_reset (ABI 4): eb_env_step resets the envs it has just finished, same call
    _fields_ = [('seed', C.c_uint64), ('counter', C.c_uint64), ('training', C.c_int32), ('ref_idx', C.c_void_p),
                ('virtual_flag', C.c_void_p), ('v_light', C.c_void_p), ('pool', EbRespawn), ('final_obs', C.c_void_p),
                # with a flow rule (ABI 5): the flow source's part of reset instead of the pool's
                ('flow_cand_len', C.c_void_p), ('flow_phase0', C.c_void_p), ('flow_random_phase', C.c_int32),
                ('flow_seed', C.c_uint64), ('flow_counter', C.c_uint64)]


class EbFlowRule(C.Structure):    # struct eb_flow_rule (ABI 4): the flow source's step as the last stage of eb_env_step
    _fields_ = [('per_route', C.c_int32), ('active', C.c_void_p), ('timer', C.c_void_p), ('emitted', C.c_void_p),
                ('sim_step', C.c_void_p), ('lane', C.c_void_p), ('period', C.c_void_p), ('v_max', C.c_void_p),
                ('dt', C.c_float), ('exit_range', C.c_float), ('accel', C.c_float), ('lane_len', C.c_float),
                ('light_cycle', C.c_int32), ('seed', C.c_uint64), ('counter', C.c_uint64), ('cand_mode', C.c_void_p),
                ('v_light', C.c_void_p)]


class EbTimeLimit(C.Structure):   # struct eb_time_limit (ABI 5): gym's TimeLimit around the registered env, README.md:55-59
    _fields_ = [('episode_step', C.c_void_p), ('max_episode_steps', C.c_int32)]


ACT_ID = {'linear': 0, None: 0, 'relu': 1, 'elu': 2, 'tanh': 3}        # EB_ACT_*
PENALTY_ID = {'veh2veh4real': 0, 'real_punish_term': 1}                # EB_PENALTY_*

_P = C.c_void_p
_I = C.c_int32

# name -> (restype, argtypes); must list every symbol include/envbuild.h declares
PROTOTYPES = {
    'eb_last_error': (C.c_char_p, []),
    'eb_abi_version': (C.c_int, []),
    'eb_backend': (C.c_char_p, []),
    'eb_create': (C.c_int, [C.POINTER(EbConfig), C.POINTER(_P)]),
    'eb_destroy': (C.c_int, [_P]),
    'eb_sync': (C.c_int, [_P]),
    'eb_set_paths': (C.c_int, [_P, _P, _P, _P, _P, _I]),
    'eb_set_veh_modes': (C.c_int, [_P, _P, _I]),
    'eb_f_xu': (C.c_int, [_P, _I, _P, _P, C.c_float, _P, _P, _P]),
    'eb_action_transform': (C.c_int, [_P, _I, _P, _P, _P]),
    'eb_compute_rewards': (C.c_int, [_P, _I, _P, _P, _P, _P, _P]),
    'eb_compute_next_obses': (C.c_int, [_P, _I, _P, _P, _P, _I, _P, _P]),
    'eb_rollout_step': (C.c_int, [_P, _I, _P, _P, _P, _I, _P, _P, _P, _P]),
    'eb_rollout_tape': (C.c_int, [_P, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P]),
    'eb_rollout_gated': (C.c_int, [_P, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P]),
    'eb_rollout_gated_blocks': (C.c_int, [_P, _I, C.POINTER(_I)]),
    'eb_gate_feed': (C.c_int, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P]),
    'eb_rollout_step_f16': (C.c_int, [_P, _I, _P, _P, _P, _I, _P, _P, _P, _P]),
    'eb_rollout_tape_f16': (C.c_int, [_P, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P]),
    'eb_episode_summary': (C.c_int, [_P, _I, _I, _P, _P, _P, _P]),
    'eb_episode_acc_bytes': (C.c_int, [_P, _I, _I, C.POINTER(C.c_int64)]),
    'eb_rollout_step_acc': (C.c_int, [_P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P]),
    'eb_episode_acc_finish': (C.c_int, [_P, _I, _I, _P, _P, _P]),
    'eb_plan_create': (C.c_int, [_P, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, C.POINTER(_P)]),
    'eb_plan_launch': (C.c_int, [_P, _P]),
    'eb_plan_destroy': (C.c_int, [_P]),
    'eb_event_create': (C.c_int, [_P, C.POINTER(_P)]),
    'eb_event_record': (C.c_int, [_P, _P]),
    'eb_event_elapsed_ms': (C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    'eb_event_destroy': (C.c_int, [_P]),
    'eb_find_closest_point': (C.c_int, [_P, _I, _P, _P, _P, _I, _I, _P, _P, _P]),
    'eb_path_points': (C.c_int, [_P, _I, _P, _P, _I, _I, _P, _P]),
    'eb_phi_diff': (C.c_int, [_P, _I, _P, _P, _P]),
    'eb_ego_predict': (C.c_int, [_P, _I, _P, _P, _P, _P]),
    'eb_tracking_error': (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    'eb_veh_predict': (C.c_int, [_P, _I, _P, _P, _P]),
    'eb_ss': (C.c_int, [_P, _I, _P, _P, _P, _I, C.c_double, _P, _P]),
    'eb_env_ego_step': (C.c_int, [_P, _I, _P, _P, _P, _P, _P]),
    'eb_get_obs': (C.c_int, [_P, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    'eb_exit_frame': (C.c_int, [_P, _I, _P, _I, _P, _P, _P]),
    'eb_judge_done': (C.c_int, [_P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    'eb_env_step': (C.c_int, [_P, _P, _I, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'eb_ego_dynamics': (C.c_int, [_P, _I, _P, _P, _P, _P]),
    'eb_env_reset': (C.c_int, [_P, _I, _P, C.c_uint64, C.c_uint64, _I, _P, _P, _P, _P, _P, _P, _P]),
    'eb_env_reset_pool': (C.c_int, [_P, _P, _I, _P, C.c_uint64, C.c_uint64, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    'eb_traffic_respawn': (C.c_int, [_P, _I, _I, _P, _P, C.c_float, C.c_float, C.c_float, C.c_uint64, C.c_uint64, _P, _P, _P, C.c_float, _P]),
    'eb_traffic_flow_reset': (C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, _I, _I,
                                        C.c_uint64, C.c_uint64, _P, _P, _P]),
    'eb_debug_set_tile': (C.c_int, [_P, _I]),
    'eb_debug_set_tape_stepwise': (C.c_int, [_P, _I]),
    'eb_debug_set_trace': (C.c_int, [_P, _P, C.c_int64]),
    'eb_debug_set_stage_paths': (C.c_int, [_P, _I]),
    'eb_debug_set_env_waves': (C.c_int, [_P, _I]),
    'eb_debug_set_scan_prefetch': (C.c_int, [_P, _I]),
    'eb_debug_set_rollout_sched': (C.c_int, [_P, _I, _I]),
    'eb_debug_rollout_plan': (C.c_int, [_P, _I, _P]),
    'eb_debug_check_grids': (C.c_int, [_P, _I, C.c_uint64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'eb_traffic_flow_step': (C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float,
                                       _I, C.c_uint64, C.c_uint64, _P, _P, _P]),
    'eb_mlp_create': (C.c_int, [C.POINTER(EbMlpConfig), C.POINTER(_P)]),
    'eb_mlp_destroy': (C.c_int, [_P]),
    'eb_mlp_set_layer': (C.c_int, [_P, _I, _P, _P]),
    'eb_mlp_set_obs_scale': (C.c_int, [_P, _P]),
    'eb_mlp_forward': (C.c_int, [_P, _I, _P, _P, _P]),
    'eb_policy_run_batch': (C.c_int, [_P, _I, _P, C.c_float, _P, _P]),
    'eb_shield_is_safe': (C.c_int, [_P, _P, _I, _P, _P, _I, _I, _I, C.c_float, _P, _P, _P, _P, _P, _P, _P]),
}


class EbError(RuntimeError):
    pass


class CApi(object):
    def __init__(self, path):
        if not os.path.isfile(path):
            raise EbError('shared library not found: %s' % path)
        self.path = path
        self.lib = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(self.lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if self.lib.eb_abi_version() != EB_ABI_VERSION:
            raise EbError('%s: ABI version %d, expected %d'
                          % (path, self.lib.eb_abi_version(), EB_ABI_VERSION))
        self.backend = self.lib.eb_backend().decode()

    def check(self, rc):
        if rc != 0:
            msg = self.lib.eb_last_error()
            msg = msg.decode() if msg else ''
            if rc == -1:
                raise ValueError('envbuild(%s): %s' % (self.backend, msg))
            raise EbError('envbuild(%s) error %d: %s' % (self.backend, rc, msg))

    def create(self, task, n_veh, n_future, mode, device=0):
        cfg = EbConfig(EB_ABI_VERSION, TASK_ID[task] if isinstance(task, str) else int(task),
                       int(n_veh), int(n_future), int(mode), int(device))
        h = _P()
        self.check(self.lib.eb_create(C.byref(cfg), C.byref(h)))
        return h

    def mlp_create_from(self, obs_dim, n_hidden, n_units, out_dim, hidden_act, out_act, layers, obs_scale=None, device=0):
        """eb_mlp_create + eb_mlp_set_layer for every (kernel [in, out], bias [out]) pair + the obs scale."""
        import numpy as np
        cfg = EbMlpConfig(EB_ABI_VERSION, int(obs_dim), int(n_hidden), int(n_units), int(out_dim),
                          ACT_ID[hidden_act], ACT_ID[out_act], int(device))
        m = _P()
        self.check(self.lib.eb_mlp_create(C.byref(cfg), C.byref(m)))
        try:
            if len(layers) != n_hidden + 1:
                raise ValueError('expected %d (kernel, bias) pairs, got %d' % (n_hidden + 1, len(layers)))
            for L, (k, b) in enumerate(layers):
                rows = obs_dim if L == 0 else n_units
                cols = out_dim if L == n_hidden else n_units
                k = np.ascontiguousarray(k, np.float32)
                b = np.ascontiguousarray(b, np.float32)
                if k.shape != (rows, cols) or b.shape != (cols,):
                    raise ValueError('layer %d: kernel %s / bias %s, expected (%d, %d) / (%d,)' % (L, k.shape, b.shape, rows, cols, cols))
                self.check(self.lib.eb_mlp_set_layer(m, L, k.ctypes.data, b.ctypes.data))
            if obs_scale is not None:
                sc = np.ascontiguousarray(obs_scale, np.float32)
                if sc.shape != (obs_dim,):
                    raise ValueError('obs_scale must have %d entries' % obs_dim)
                self.check(self.lib.eb_mlp_set_obs_scale(m, sc.ctypes.data))
        except Exception:
            self.lib.eb_mlp_destroy(m)
            raise
        return m

    def __getattr__(self, name):
        # eb_xxx(...) with return-code checking: api.rollout_step(h, ...)
        fn = getattr(self.lib, 'eb_' + name)

        def call(*args):
            self.check(fn(*args))
        call.__name__ = name
        setattr(self, name, call)
        return call


HIP_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib',
                            'libenvbuild_hip.so')
_hip_api = None


def hip_api():
    """The product's only backend.  Raises (never falls back) when the HIP library is absent."""
    global _hip_api
    if _hip_api is None:
        # PyTorch-ROCm bundles its own libamdhip64; import it FIRST so that this library binds to the
        # same HIP runtime (one runtime per process: shared device pointers and streams).  Loading
        # ours first makes torch see no GPU.
        import torch  # noqa: F401
        from . import build as _build
        ok, why = _build.check_fresh()
        if not ok:
            # never load a binary that does not come from the sources next to it: rebuild when the compiler is
            # here (seconds), refuse otherwise
            hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
            if os.path.isfile(hipcc):
                try:
                    _build.build()
                except Exception as e:   # noqa: BLE001
                    raise EbError('%s; rebuilding it failed: %s.  env_build_amd has no CPU fallback.' % (why, e))
            else:
                raise EbError('%s — run `python -c "import __graft_entry__ as g; g.build()"` (hipcc '
                              '--offload-arch=gfx950).  env_build_amd has no CPU fallback.' % why)
        _hip_api = CApi(HIP_LIB_PATH)
        if _hip_api.backend != 'hip':
            raise EbError('%s is not the HIP backend' % HIP_LIB_PATH)
    return _hip_api
