"""Drop-in counterpart of the reference's endtoend.py Gym env — CrossroadEnd2end — with the arithmetic
of step / reset running in HIP kernels on an MI355X (E2E = /root/reference/endtoend.py).

Same constructor arguments, methods and attributes as the reference (E2E:44-507): `reset(**kwargs) ->
obs`, `step(action) -> (obs, reward, done, info)`, `seed`, `close`, `set_traj`, `_get_obs(exit_)`,
`compute_reward(obs, action)`, `action_space`, `obs`, `action`, `ego_dynamics`, `all_vehicles`, `v_light`,
`done_type`, `reward_info`, `ref_path`, `env_model`, `dynamics`, `veh_num`.  Two additions:

  * `n_env` (default 1): a BATCH of independent single-ego envs advanced by one set of kernel launches
    per step.  With n_env == 1 every return value has the reference's shape and type (obs float32 [D],
    reward np.float32, done int, info dict); with n_env > 1 they are device arrays of leading size B
    (obs [B, D], reward [B], done uint8 [B]) and `done_type` holds the uint8 done codes (`done_names()` spells them).
  * the traffic source.  The reference co-simulates with SUMO over TraCI (traffic.py; out of scope,
    SURVEY.md §2 #6).  Here the surrounding vehicles are vehicle slots per env advanced with the model's own
    prediction step (EnvironmentModel.veh_predict, DAM:394-427 — the same arithmetic the reference's 5/20-step
    safety shield trusts).  `traffic='pool'` (default): a fixed pool of `n_cand` candidates re-entered at
    their lane's start when they leave the map.  `traffic='flows'`: the twelve flows of sumo_files/cross.rou.xml
    (emission periods, vTypes, random depart position / speed), the traffic-light programme and
    init_traffic's conflict removal — env_build_amd/traffic.py.  As in the reference's `multi_display=True`
    mode (multi_ego.py:46-48, 94-96), `all_vehicles`, `ego_dynamics` and `v_light` can also be injected by
    hand before `_get_obs(exit_)`.

Per step (E2E:132-144): action scaling (E2E:133) -> reward on the CURRENT obs (E2E:134, DAM:186-320) ->
ego bicycle-model step, v_x floored at 0, phi wrapped (E2E:135, 269-283) -> traffic step -> observation
= ego | tracking error | filtered / sorted / padded vehicles (E2E:140, 285-464) -> done code by the
reference's priority collision > road > deviation > stability > red light > goal (E2E:141, 200-256,
TRF:263-295, UTL:73-104).  Every one of these is a C-ABI call into libenvbuild_hip.so; torch only owns
the device buffers.  There is no CPU path.
"""
import ctypes as C
import weakref

import numpy as np
import torch

from . import _capi
from .dynamics_and_models import (DevArray, EnvironmentModel, ReferencePath, VehicleDynamics, _Handle, _resolve_device,
                                  _dev, _ptr, _stream)
from .endtoend_env_utils import (CROSSROAD_SIZE, EXPECTED_V, L, LANE_NUMBER, LANE_WIDTH, VEH_NUM, VEHICLE_MODE_DICT,
                                 VEHICLE_MODE_LIST, W, rotate_and_shift_coordination)

__all__ = ['CrossroadEnd2end', 'Box']


class Box(object):
    """The slice of gym.spaces.Box the reference's callers use (E2E:64, 89)."""

    def __init__(self, low, high, shape, dtype=np.float32, seed=None):
        self.low, self.high, self.shape, self.dtype = float(low), float(high), tuple(shape), dtype
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high, self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


# exit-relative edge names of the reference's route classifier (E2E:345-348)
_NAME_SETTINGS = dict(D=dict(do='1o', di='1i', ro='2o', ri='2i', uo='3o', ui='3i', lo='4o', li='4i'),
                      R=dict(do='2o', di='2i', ro='3o', ri='3i', uo='4o', ui='4i', lo='1o', li='1i'),
                      U=dict(do='3o', di='3i', ro='4o', ri='4i', uo='1o', ui='1i', lo='2o', li='2i'),
                      L=dict(do='4o', di='4i', ro='1o', ri='1i', uo='2o', ui='2i', lo='3o', li='3i'))
_MODE_EDGES = dict(dl=('do', 'li'), du=('do', 'ui'), dr=('do', 'ri'), rd=('ro', 'di'), rl=('ro', 'li'), ru=('ro', 'ui'),
                   ur=('uo', 'ri'), ud=('uo', 'di'), ul=('uo', 'li'), lu=('lo', 'ui'), lr=('lo', 'ri'), ld=('lo', 'di'))


def classify_route(route, exit_='D'):
    """(start, end) edge names -> vehicle mode under the exit-relative naming (E2E:352-385), or None."""
    names = _NAME_SETTINGS[exit_]
    for mode, (s, e) in _MODE_EDGES.items():
        if route[0] == names[s] and route[1] == names[e]:
            return mode
    return None


def _lane_entry(mode):
    """(x, y, phi, along-lane unit vector) where a vehicle of `mode` enters the map: the lanes the fill
    values of E2E:439-447 are parked on."""
    h = CROSSROAD_SIZE / 2
    lane = {'l': 0.5, 'u': 1.5, 'r': LANE_NUMBER - 0.5}
    s, e = mode[0], mode[1]
    if s == 'd':
        return LANE_WIDTH * lane[e], -(h + 35.), 90., (0., 1.)
    if s == 'u':
        off = {'r': 0.5, 'd': 1.5, 'l': LANE_NUMBER - 0.5}[e]
        return -LANE_WIDTH * off, (h + 35.), -90., (0., -1.)
    if s == 'r':
        off = {'d': 0.5, 'l': 1.5, 'u': LANE_NUMBER - 0.5}[e]
        return (h + 35.), LANE_WIDTH * off, 180., (-1., 0.)
    off = {'u': 0.5, 'r': 1.5, 'd': LANE_NUMBER - 0.5}[e]
    return -(h + 35.), -LANE_WIDTH * off, 0., (1., 0.)


def _unwrap_action(action, B, dev):
    """[B, 2] float32 contiguous tensor on `dev` from whatever the caller passed (no copy when it already is one)."""
    t = action.t if isinstance(action, DevArray) else action
    if isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.device == dev and t.is_contiguous() and t.shape == (B, 2):
        return t
    if isinstance(t, torch.Tensor):
        return t.to(device=dev, dtype=torch.float32).reshape(B, 2).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(t, np.float32).reshape(B, 2))).to(dev)


class _RewardInfo(dict):
    """info['reward_info'] of a batch (E2E:142-143): the 16 reward terms are rows of one [16, B] device array; a row is
    wrapped as a DevArray when it is first read, so a driver that never looks pays nothing per step.  The array itself is
    optional: until somebody has read a term once, step() does not ask the kernel for it (64 B per env-step) and `compute`
    makes it on demand from the observation and action the step consumed (eb_compute_rewards: the same bits)."""

    _ROWS = {}                               # keys tuple -> {key: row}; built once, shared by every step's dict

    def __init__(self, keys, d16, final_rew, compute=None):
        dict.__init__(self)
        rows = self._ROWS.get(keys)
        if rows is None:
            rows = self._ROWS[keys] = {k: i for i, k in enumerate(keys)}
        self._rows = rows
        self._d16 = d16
        self._compute = compute
        dict.__setitem__(self, 'final_rew', final_rew)

    def __missing__(self, k):
        row = self._rows[k]                  # KeyError for a name that is not a reward term
        if self._d16 is None:
            self._d16 = self._compute()
            self._compute = None
        v = DevArray(self._d16[row])
        dict.__setitem__(self, k, v)
        return v

    def _fill(self):
        for k in self._rows:
            self[k]

    def keys(self):
        self._fill(); return dict.keys(self)

    def items(self):
        self._fill(); return dict.items(self)

    def values(self):
        self._fill(); return dict.values(self)

    def get(self, k, default=None):          # (dict.get bypasses __missing__)
        try:
            return self[k]
        except KeyError:
            return default

    def copy(self):
        self._fill(); return dict(dict.items(self))

    def __reduce__(self):                    # pickles / deep-copies as the plain dict it stands for
        self._fill(); return (dict, (dict(dict.items(self)),))

    def __iter__(self):
        self._fill(); return dict.__iter__(self)

    def __len__(self):
        return len(self._rows) + 1

    def __contains__(self, k):
        return k in self._rows or dict.__contains__(self, k)


class _LazyInitState(dict):
    """init_state of a batch (E2E:487-499 builds it for the one env of the reference): env 0's start values, read from the
    device when somebody asks — a masked reset of a vectorised driver must not sync the stream for a dict nobody reads."""

    def __init__(self, ego, l, w, route):
        dict.__init__(self)
        self._src = (ego, l, w, route)

    def __missing__(self, k):
        if k != 'ego':
            raise KeyError(k)
        ego, l, w, route = self._src
        e0 = ego[0].cpu().numpy()
        v = dict(v_x=e0[0], v_y=0, r=0, x=e0[3], y=e0[4], phi=e0[5], l=l, w=w, routeID=route)
        dict.__setitem__(self, k, v)
        return v

    def __contains__(self, k):
        return k == 'ego'

    def get(self, k, default=None):
        return self[k] if k == 'ego' else default

    def keys(self):
        self['ego']; return dict.keys(self)

    def items(self):
        self['ego']; return dict.items(self)

    def values(self):
        self['ego']; return dict.values(self)

    def __iter__(self):
        self['ego']; return dict.__iter__(self)

    def __len__(self):
        return 1

    def copy(self):
        self['ego']; return dict(dict.items(self))

    def __reduce__(self):
        self['ego']; return (dict, (dict(dict.items(self)),))


_TIME_LIMIT_CODE = _capi.DONE_NAMES.index('time_limit')     # EB_DONE_TIME_LIMIT
MAX_EPISODE_STEPS = 200     # the reference's registration of 'CrossroadEnd2end-v0' (README.md:55-59)


class _LazyDone(DevArray):
    """done of a batch: uint8 0 / 1 per env = (done code != 0) — the one small kernel that makes it runs when the value
    is first read (`.t`, `.numpy()`, indexing, reset(mask=done), ...), not on every step."""

    def __init__(self, code):
        self._code = code
        self._t = None

    @property
    def t(self):
        if self._t is None:
            self._t = self._code.clamp(max=1)
        return self._t

    @t.setter
    def t(self, v):
        self._t = v


class _LazyTruncated(DevArray):
    """info['TimeLimit.truncated'] of a batch: uint8 0 / 1 per env = (done code == EB_DONE_TIME_LIMIT), made when first read"""

    def __init__(self, code):
        self._code = code
        self._t = None

    @property
    def t(self):
        if self._t is None:
            self._t = (self._code == _TIME_LIMIT_CODE).to(torch.uint8)
        return self._t

    @t.setter
    def t(self, v):
        self._t = v


class _SnapshotOnDemand(DevArray):
    """A state array that the NEXT step / reset overwrites in place (info['ref_index'] under auto_reset), handed out without a
    copy: the copy is made when the value is first read or — if the object is still referenced by then — right before the array
    is written again (CrossroadEnd2end._settle_snapshots).  A driver that drops `info` every step pays nothing."""

    def __init__(self, src):
        self._src = src
        self._t = None

    @property
    def t(self):
        if self._t is None:
            self._t = self._src.clone()
            self._src = None
        return self._t

    @t.setter
    def t(self, v):
        self._t = v


class CrossroadEnd2end(object):
    def __init__(self, training_task, num_future_data=0, mode='training', multi_display=False, n_env=1, n_cand=None,
                 device=None, respawn=True, traffic='pool', per_route=5, auto_reset=False, copy_outputs=True, flow_in_step=True,
                 max_episode_steps=None, **kwargs):
        """n_env > 1 makes a batch of independent single-ego envs (the reference is one env: every argument before n_env is its).
        max_episode_steps: None — the bare class, as `CrossroadEnd2end(...)` in the reference; an int — the env as its callers get it
            from gym.make('CrossroadEnd2end-v0') (README.md:55-59 registers it with max_episode_steps = 200; `make()` below), i.e.
            inside gym's TimeLimit: an episode nothing else has ended ends at that many steps with done_type 'time_limit' and
            info['TimeLimit.truncated'] set; the count lives on the device, is kept by the step's own kernel launch
            (eb_time_limit) and restarts with every reset.
        auto_reset (a batch): step() also resets the envs it has just finished — over the traffic pool in the same kernel launch
            (eb_auto_reset), over the flow source as the masked reset's launches behind the step's —
            the observation it returns holds their reset observation, info['final_observation'] their terminal one (rows of the
            other envs unspecified), `done` says which.  The vectorised-env convention; hier_decision.py:109-135's loop in one call.
        copy_outputs: True (default) — what step() / reset() hand out are arrays of their own, as in the reference: they can be
            kept in a rollout list or a replay buffer.  False — the outputs live in two pre-allocated buffer sets used in turn
            (zero allocations per step): a value handed out stays valid until the step AFTER the next one and must be consumed or
            copied by then; `done` and info['reward_info'] are materialised on first read and must be read in that window too."""
        if training_task not in ('left', 'straight', 'right'):
            raise ValueError("training_task must be 'left', 'straight' or 'right'")
        self.device = _resolve_device(device)
        self.n_env = int(n_env)
        if self.n_env < 1:
            raise ValueError('n_env must be >= 1')
        self.dynamics = VehicleDynamics(self.device)                                   # E2E:51
        self.interested_vehs = None
        self.training_task = training_task
        self.ref_path = ReferencePath(self.training_task, device=self.device, **kwargs)   # E2E:54
        self.detected_vehicles = None
        self.all_vehicles = None
        self.ego_dynamics = None
        self.num_future_data = num_future_data
        self.env_model = EnvironmentModel(training_task, num_future_data, device=self.device)   # E2E:59
        self.init_state = {}
        self.action_number = 2
        self.exp_v = EXPECTED_V
        self.ego_l, self.ego_w = L, W
        self.action_space = Box(low=-1, high=1, shape=(self.action_number,), dtype=np.float32)   # E2E:64
        self.seed()
        self.v_light = 0
        self.step_length = 100  # ms
        self.step_time = self.step_length / 1000.0
        self.obs = None
        self.action = None
        self.veh_mode_dict = VEHICLE_MODE_DICT[self.training_task]
        self.veh_num = VEH_NUM[self.training_task]
        self.virtual_red_light_vehicle = False
        self.done_type = 'not_done_yet'
        self.reward_info = None
        self.ego_info_dim = 6
        self.per_tracking_info_dim = 3
        self.per_veh_info_dim = 4
        self.mode = mode
        self.multi_display = multi_display
        self.respawn = respawn
        self.copy_outputs = bool(copy_outputs)
        self.flow_in_step = bool(flow_in_step)   # traffic='flows': the source's step inside the env step's launch (eb_flow_rule), or a launch of its own
        self.auto_reset = bool(auto_reset)
        self._want_d16 = False     # the 16-term reward dict is requested from the kernel once somebody has read a term
        self.obs_dim = 6 + 3 * (num_future_data + 1) + 4 * self.veh_num
        self.observation_space = Box(-np.inf, np.inf, (self.obs_dim,), np.float32)

        # handles: the model's (native slot list: rewards, obs construction, done) and one over the candidate pool
        self.api, self._h = self.env_model.api, self.env_model.handle
        native = VEHICLE_MODE_LIST[self.training_task]
        if traffic not in ('pool', 'flows'):
            raise ValueError("traffic must be 'pool' or 'flows'")
        self.traffic_kind = traffic
        self.cand_modes = list(native) * 2 if n_cand is None else [native[i % len(native)] for i in range(int(n_cand))]
        if traffic == 'flows':      # twelve SUMO flows, per_route slots each (env_build_amd/traffic.py)
            from .traffic import ROUTES
            self.cand_modes = [r for r in ROUTES for _ in range(int(per_route))]
        self.n_cand = len(self.cand_modes)
        if not 1 <= self.n_cand <= 64:
            raise ValueError('n_cand must be in 1..64')
        self._traffic = _Handle(self.training_task, self.n_cand, 0, _capi.MODE_SELECTING, self.device,
                                modes=self.cand_modes, with_paths=False)
        B, M, dev = self.n_env, self.n_cand, self.device
        self._ego = torch.zeros((B, 6), dtype=torch.float32, device=dev)
        self._params = torch.zeros((B, 4), dtype=torch.float32, device=dev)
        self._cand = torch.zeros((B, M, 4), dtype=torch.float32, device=dev)
        self._cand_mode = torch.tensor([_capi.VMODE_ID[m] for m in self.cand_modes], dtype=torch.uint8,
                                       device=dev).repeat(B, 1).contiguous()
        self._v_light = torch.zeros((B,), dtype=torch.uint8, device=dev)
        self._virtual = torch.zeros((B,), dtype=torch.uint8, device=dev)
        self._ref_idx = torch.zeros((B,), dtype=torch.int32, device=dev)
        self._obs = torch.zeros((B, self.obs_dim), dtype=torch.float32, device=dev)
        self._entry = torch.tensor([_lane_entry(m)[:3] for m in self.cand_modes], dtype=torch.float32, device=dev)
        self._entry_dir = torch.tensor([_lane_entry(m)[3] for m in self.cand_modes], dtype=torch.float32, device=dev)
        self._entry5 = torch.cat([self._entry, self._entry_dir], 1).contiguous()       # (x, y, phi, dx, dy) per slot
        self._virtual_next = torch.zeros((B,), dtype=torch.uint8, device=dev)
        self._exit_id = None
        self._ego_exit = None
        self._reset_counter = 0
        self.done_code = torch.zeros((B,), dtype=torch.uint8, device=dev)
        self.max_episode_steps = None if max_episode_steps is None else int(max_episode_steps)
        if self.max_episode_steps is not None and self.max_episode_steps < 1:
            raise ValueError('max_episode_steps must be >= 1')
        self._episode_step = torch.zeros((B,), dtype=torch.int32, device=dev)     # gym's TimeLimit._elapsed_steps, per env
        self._time_limit = None if self.max_episode_steps is None else \
            _capi.EbTimeLimit(self._episode_step.data_ptr(), self.max_episode_steps)
        self._injected = False
        self._flows = None
        self._bufs, self._buf_i, self._ri1 = None, 0, None
        self._ri_snapshot = None                     # weakref to the info['ref_index'] handed out last (copy_outputs)
        self._rbufs, self._rbuf_i = None, 0      # reset(mask=...) over the pool: its observation / done-code sets
        # during an episode a vehicle that left the map re-enters at its lane's edge (within POOL_EDGE_SPAN m of the entry
        # point, 60 m from the centre: where no ego is), not somewhere along the lane
        self._respawn_rule = _capi.EbRespawn(self._entry5.data_ptr(), CROSSROAD_SIZE / 2 + 40., self.POOL_EDGE_SPAN, EXPECTED_V, 0, 0)
        # a reset spreads the pool over the first 60 m of every lane, clear of the ego (eb_env_reset_pool)
        self._reset_rule = _capi.EbRespawn(self._entry5.data_ptr(), 0.0, 60.0, EXPECTED_V, 0, 0, self.POOL_EDGE_SPAN)
        if self.auto_reset and (B == 1 or traffic not in ('pool', 'flows')):
            raise ValueError('auto_reset needs a batch (n_env > 1) over the traffic pool or the flow source')
        self._auto_rule = None
        # the flow source with its rule as a launch of its own (flow_in_step=False): step() composes the same contract from the step
        # launch and the masked reset's launches; with the rule inside the step's launch the reset rides there too (ABI 5)
        self._auto_compose = self.auto_reset and traffic == 'flows' and not self.flow_in_step
        if self.auto_reset and traffic == 'pool':       # eb_auto_reset: the state arrays are fixed, the counters and final_obs are set per step
            self._auto_rule = _capi.EbAutoReset(0, 0, 1 if self.mode == 'training' else 0, self._ref_idx.data_ptr(),
                                                self._virtual.data_ptr(), self._v_light.data_ptr(), self._reset_rule, None)
        if traffic == 'flows':
            from .traffic import FlowTraffic
            self._flows = FlowTraffic(B, dev, None, self.training_task, mode=self.mode, per_route=per_route,
                                      step_time=self.step_time)
            self._flows.seed = self._respawn_seed
            if self.auto_reset and self.flow_in_step:   # eb_auto_reset over the flow source: the source's own arrays (its light is the env's)
                fl = self._flows
                self._auto_rule = _capi.EbAutoReset(0, 0, 1 if self.mode == 'training' else 0, self._ref_idx.data_ptr(),
                                                    self._virtual.data_ptr(), fl.v_light().data_ptr(), self._reset_rule, None,
                                                    fl.veh_len.data_ptr(), fl.phase0.data_ptr(),
                                                    1 if self.training_task == 'right' else 0, 0, 0)
        self.init_state = self._reset_init_state()
        if not multi_display:                                                           # E2E:84-93
            self.reset()
            self.step(self.action_space.sample() if B == 1 else
                      np.stack([self.action_space.sample() for _ in range(B)]))
            self.reset()

    # -- gym plumbing ---------------------------------------------------------------------------
    def seed(self, seed=None):  # E2E:95-97
        self.np_random = np.random.default_rng(seed)
        # every device-side draw is counter-based (eb_env_reset, eb_traffic_respawn, eb_traffic_flow_*): one 62-bit seed
        self._respawn_seed = int(self.np_random.integers(0, 2 ** 62))     # keys: (seed, counter, env, slot)
        if getattr(self, '_flows', None) is not None:
            self._flows.seed = self._respawn_seed
            self._flows.counter = self._flows.reset_counter = 0      # a reseed reproduces the same flow traffic
        self._respawn_counter = 0
        self._reset_counter = 0
        return [seed]

    def close(self):  # E2E:129-130
        self._traffic = None

    def set_traj(self, trajectory):  # E2E:793-795
        self.ref_path = trajectory

    def render(self, mode='human'):  # E2E:509-791 is a matplotlib view: out of scope
        raise NotImplementedError('CrossroadEnd2end.render (matplotlib view) is out of scope')

    # -- helpers --------------------------------------------------------------------------------
    def _sp(self):
        return _stream(self.device)

    def _ret(self, t):
        """reference-shaped value for n_env == 1, device array otherwise"""
        return t[0].detach().cpu().numpy() if self.n_env == 1 else DevArray(t)

    # -- reset ----------------------------------------------------------------------------------
    _RESET_SALT, _POOL_SALT = 0x2545F4914F6CDD1D, 0x5DEECE66D
    POOL_EDGE_SPAN = 5.0      # metres from the map edge within which the pool re-enters a vehicle during an episode / on a conflict

    def _reset_init_state(self, mask8=None):  # E2E:472-499, per env
        """n_env == 1: the reference's own host-side draws (np.random, E2E:474-482).  A batch: ONE kernel, eb_env_reset —
        path, start index, start speed, parameters, the next virtual-red-light flag and the done code of the masked envs."""
        span = {'left': 900 + 500, 'straight': 1200 + 500, 'right': 420 + 500}[self.training_task]
        B, dev = self.n_env, self.device
        route = {'left': 'dl', 'straight': 'du', 'right': 'dr'}[self.training_task]
        miu = self.dynamics.vehicle_params['miu']
        if B == 1:
            ref = int(self.ref_path.ref_index)
            index = int(self.np_random.random() * span) + 700                            # E2E:474-478
            v = np.float32(EXPECTED_V * self.np_random.random())                         # E2E:482
            path = self.ref_path.path_list[ref]
            i = int(np.clip(index, 0, len(path[0]) - 1))                                 # indexs2points, DAM:727-728
            ego = np.array([[v, 0., 0., path[0][i], path[1][i], path[2][i]]], np.float32)
            self._ego.copy_(torch.from_numpy(ego))
            self._params.copy_(torch.tensor([[0., 0., miu, miu]], dtype=torch.float32))  # E2E:110-113
            self._ref_idx.fill_(ref)
            if self._time_limit is not None:
                self._episode_step.zero_()                                               # TimeLimit.reset
            e0 = ego[0]
        else:
            self._reset_counter += 1
            self.api.env_reset(self._h, B, _ptr(mask8), C.c_uint64(self._respawn_seed ^ self._RESET_SALT),
                               C.c_uint64(self._reset_counter), 1 if self.mode == 'training' else 0, _ptr(self._ego),
                               _ptr(self._params), _ptr(self._ref_idx), _ptr(self._virtual_next), _ptr(self.done_code),
                               _ptr(self._episode_step) if self._time_limit is not None else None, self._sp())
            return _LazyInitState(self._ego, self.ego_l, self.ego_w, route)     # env 0's values, copied from the device on first access
        return dict(ego=dict(v_x=e0[0], v_y=0, r=0, x=e0[3], y=e0[4], phi=e0[5], l=self.ego_l,
                             w=self.ego_w, routeID=route))

    def reset(self, **kwargs):  # E2E:99-127
        """`mask=` (n_env > 1; bool / uint8 [B]) resets only those envs — the vectorised-env idiom for batched drivers."""
        mask = kwargs.pop('mask', None)
        if self._ri_snapshot is not None:
            self._settle_snapshots()                 # (a reset rewrites `_ref_idx` in place)
        if kwargs or self.ref_path is None:
            self.ref_path = ReferencePath(self.training_task, device=self.device, **kwargs)
        elif self.n_env == 1:
            self.ref_path = ReferencePath(self.training_task, device=self.device)       # E2E:100: a fresh random path
        B, dev = self.n_env, self.device
        mask8 = None
        if mask is not None and B > 1:
            if isinstance(mask, _LazyDone) and mask._t is None:
                mt = mask._code          # step()'s `done`, not read yet: its done codes serve as the mask (non-zero = reset)
            else:
                mt = mask.t if isinstance(mask, DevArray) else mask
            mt = mt if isinstance(mt, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(mt)))
            mask8 = mt.to(device=dev).reshape(B).to(torch.uint8).contiguous()
        sp = self._sp()
        if B > 1 and self._flows is None and self._exit_id is None:
            # the whole masked reset over the traffic pool as ONE C call and one launch (eb_env_reset_pool): state and flags
            # (E2E:100-101, 119), the pool's re-entry clear of the ego (E2E:102-103), v_light cleared, the reset observation of
            # those rows with the OLD flags (E2E:116), the drawn flags swapped in (E2E:120-126).  The observation and the done
            # codes go to one of two pre-allocated sets used in turn; the rows of the other envs are carried over inside the
            # kernel (obs_src / done_src), so the arrays the last step handed out stay as they were and nothing is cloned.
            self._reset_counter += 2
            if not self._cand.is_contiguous():
                self._cand = self._cand.contiguous()
            if self.copy_outputs:                                     # arrays of their own, every call
                rb = dict(obs=torch.empty((B, self.obs_dim), dtype=torch.float32, device=dev),
                          code=torch.empty((B,), dtype=torch.uint8, device=dev))
            else:
                if self._rbufs is None:
                    self._rbufs = [dict(obs=torch.empty((B, self.obs_dim), dtype=torch.float32, device=dev),
                                        code=torch.empty((B,), dtype=torch.uint8, device=dev)) for _ in range(2)]
                self._rbuf_i ^= 1
                rb = self._rbufs[self._rbuf_i]
                if rb['obs'].data_ptr() == self._obs.data_ptr():      # two resets in a row: the other set
                    self._rbuf_i ^= 1
                    rb = self._rbufs[self._rbuf_i]
            rule = self._reset_rule
            rule.seed, rule.counter = self._respawn_seed ^ self._POOL_SALT, self._reset_counter
            if self._auto_rule is not None:
                rule = self._auto_rule.pool                           # (the struct holds its own copy of the pool rule)
                rule.seed, rule.counter = self._respawn_seed ^ self._POOL_SALT, self._reset_counter
            self.api.env_reset_pool(self._h, self._traffic.h, B, _ptr(mask8), C.c_uint64(self._respawn_seed ^ self._RESET_SALT),
                                    C.c_uint64(self._reset_counter - 1), 1 if self.mode == 'training' else 0, _ptr(self._ego),
                                    _ptr(self._params), _ptr(self._ref_idx), _ptr(self._virtual), _ptr(self._v_light),
                                    _ptr(rb['code']), _ptr(self._episode_step) if self._time_limit is not None else None,
                                    self.n_cand, _ptr(self._cand), _ptr(self._cand_mode), C.byref(rule),
                                    _ptr(rb['obs']), _ptr(self._obs), _ptr(self.done_code), sp)
            self._obs, self.done_code = rb['obs'], rb['code']
            route = {'left': 'dl', 'straight': 'du', 'right': 'dr'}[self.training_task]
            self.init_state = _LazyInitState(self._ego, self.ego_l, self.ego_w, route)
            self._injected = False
            self.ego_dynamics = self.all_vehicles = None
            self.obs = DevArray(self._obs)
            self.action = None
            self.reward_info = None
            self.virtual_red_light_vehicle = None
            self.done_type = DevArray(self.done_code)
            return self.obs
        if B > 1:
            self.done_code = self.done_code.clone()      # the array handed out by the last step stays as it was
        self.init_state = self._reset_init_state(mask8)                                 # E2E:101
        if self._flows is not None:                                                      # E2E:102-103 (init_traffic)
            self._flows.reset(self.api, self._traffic.h, mask8, self._ego, sp)
            self._cand, self._cand_mode, self._v_light = self._flows.cand, self._flows.mode(), self._flows.v_light()
        else:   # the pool: every candidate of the chosen envs re-enters on its lane (limit < 0 = unconditional)
            self._reset_counter += 1
            if not self._cand.is_contiguous():
                self._cand = self._cand.contiguous()
            # (init_traffic's conflict rule, TRF:168-192: a candidate that would start on top of the ego goes to its lane's edge)
            self.api.traffic_respawn(self._traffic.h, B, self.n_cand, _ptr(self._cand), _ptr(self._entry5), C.c_float(-1.0),
                                     C.c_float(60.0), C.c_float(EXPECTED_V), C.c_uint64(self._respawn_seed ^ self._POOL_SALT),
                                     C.c_uint64(self._reset_counter), _ptr(mask8), None, _ptr(self._ego), C.c_float(self.POOL_EDGE_SPAN), sp)
            if mask8 is None:                      # the pool has no light programme: a reset env starts at phase 0
                self._v_light.zero_()              # (a value injected through the multi_display seam does not survive reset)
            else:
                self._v_light.masked_fill_(mask8 != 0, 0)
        self._injected = False
        self._publish_state()                                                            # E2E:104-115
        if B > 1:      # eb_get_obs writes the (masked) rows in place: not into the array the last step handed out
            self._obs = self._obs.clone() if mask8 is not None else torch.empty_like(self._obs)
        self.obs = self._get_obs(row_mask=mask8)                                         # E2E:116 (with the OLD flag)
        self.action = None
        self.reward_info = None
        if B == 1:
            self.done_type = 'not_done_yet'
            flag = bool(self.mode == 'training' and self.np_random.random() > 0.9)       # E2E:120-126
            self._virtual.fill_(1 if flag else 0)
            self.virtual_red_light_vehicle = flag
        else:
            # E2E:120-126: the flag drawn by eb_env_reset replaces the old one AFTER the reset observation, for the reset envs only
            if mask8 is None:
                self._virtual.copy_(self._virtual_next)
            else:
                torch.where(mask8 != 0, self._virtual_next, self._virtual, out=self._virtual)
            self.virtual_red_light_vehicle = None
            self.done_type = DevArray(self.done_code)
        return self.obs

    # -- reference-shaped views of the device state (n_env == 1) ---------------------------------
    def _publish_state(self):
        if self.n_env != 1:
            self.ego_dynamics = self.all_vehicles = None
            return
        ego = self._ego[0].cpu().numpy()
        par = self._params[0].cpu().numpy()
        self.ego_dynamics = self._get_ego_dynamics(ego, par)
        cand = self._cand[0].cpu().numpy()
        names = _NAME_SETTINGS['D']
        self.all_vehicles = [dict(x=float(c[0]), y=float(c[1]), v=float(c[2]), phi=float(c[3]), l=L, w=W,
                                  route=(names[_MODE_EDGES[m][0]], names[_MODE_EDGES[m][1]]))
                             for c, m, cm in zip(cand, self.cand_modes, self._cand_mode[0].cpu().numpy()) if cm != _capi.VMODE_EMPTY]
        self.v_light = int(self._v_light[0].item())

    def _get_ego_dynamics(self, next_ego_state, next_ego_params):  # E2E:150-183 (host floats, as in the reference)
        out = dict(v_x=next_ego_state[0], v_y=next_ego_state[1], r=next_ego_state[2], x=next_ego_state[3],
                   y=next_ego_state[4], phi=next_ego_state[5], l=self.ego_l, w=self.ego_w,
                   alpha_f=next_ego_params[0], alpha_r=next_ego_params[1], miu_f=next_ego_params[2],
                   miu_r=next_ego_params[3])
        p = self.dynamics.vehicle_params
        alpha_f_bound, alpha_r_bound = 3 * out['miu_f'] * p['F_zf'] / p['C_f'], 3 * out['miu_r'] * p['F_zr'] / p['C_r']
        r_bound = out['miu_r'] * p['g'] / (abs(out['v_x']) + 1e-8)
        l, w, x, y, phi = out['l'], out['w'], out['x'], out['y'], out['phi']
        corners = tuple(rotate_and_shift_coordination(sx * l / 2, sy * w / 2, 0, -x, -y, -phi)[:2]
                        for sx, sy in ((1, 1), (1, -1), (-1, 1), (-1, -1)))
        out.update(dict(alpha_f_bound=alpha_f_bound, alpha_r_bound=alpha_r_bound, r_bound=r_bound, Corner_point=corners))
        return out

    def ego_dynamics_batch(self):
        """_get_ego_dynamics' derived entries for every env of the batch (eb_ego_dynamics, E2E:150-183) -> dict of DevArrays:
        alpha_f_bound, alpha_r_bound, r_bound [B] and Corner_point [B, 4, 2] — the values the done judge decides on."""
        out = torch.empty((self.n_env, 11), dtype=torch.float32, device=self.device)
        self.api.ego_dynamics(self._h, self.n_env, _ptr(self._ego), _ptr(self._params), _ptr(out), self._sp())
        return dict(alpha_f_bound=DevArray(out[:, 0]), alpha_r_bound=DevArray(out[:, 1]), r_bound=DevArray(out[:, 2]),
                    Corner_point=DevArray(out[:, 3:].reshape(self.n_env, 4, 2)))

    def _absorb_injected(self, exit_):
        """multi_display seam (multi_ego.py:94-96): the caller assigned ego_dynamics / all_vehicles / v_light."""
        if self.n_env != 1 or self.ego_dynamics is None or self.all_vehicles is None:
            return None, None
        ed = self.ego_dynamics
        ego = torch.tensor([[ed['v_x'], ed['v_y'], ed['r'], ed['x'], ed['y'], ed['phi']]], dtype=torch.float32)
        vs = [(v, classify_route(v['route'], exit_)) for v in self.all_vehicles if v.get('route') is not None]
        vs = [(v, m) for v, m in vs if m is not None]
        cand = torch.tensor([[v['x'], v['y'], v['v'], v['phi']] for v, _ in vs], dtype=torch.float32).reshape(1, -1, 4)
        cmode = torch.tensor([[_capi.VMODE_ID[m] for _, m in vs]], dtype=torch.uint8).reshape(1, -1)
        self._ego.copy_(ego)
        self._v_light.fill_(1 if self.v_light else 0)
        return cand.to(self.device), cmode.to(self.device)

    # -- observation (E2E:285-303, 329-464) -------------------------------------------------------
    def _exit_ids(self, exit_):
        """'D' / 'R' / 'U' / 'L' for every env, or one id (or letter) per env -> uint8 [B] on the device"""
        if isinstance(exit_, str):
            return torch.full((self.n_env,), _capi.EXIT_ID[exit_], dtype=torch.uint8, device=self.device)
        t = exit_.t if isinstance(exit_, DevArray) else exit_
        if not isinstance(t, torch.Tensor):
            a = np.asarray(t)
            if a.dtype.kind in 'US':
                a = np.array([_capi.EXIT_ID[str(e)] for e in a.ravel()], np.uint8)
            t = torch.from_numpy(np.ascontiguousarray(a.astype(np.uint8)))
        return t.to(device=self.device, dtype=torch.uint8).reshape(self.n_env).contiguous()

    def exit_frame(self, ego, exit_, inverse=False):
        """cal_ego_info_in_transform_coordination for a batch (UTL:184-196, multi_ego.py:88, 118): ego [B, 6] into
        (or, inverse=True, back out of) the frame of each env's exit.  One kernel (eb_exit_frame)."""
        e = _dev(ego, self.device).reshape(self.n_env, 6).contiguous()
        out = torch.empty_like(e)
        self.api.exit_frame(self._h, self.n_env, _ptr(self._exit_ids(exit_)), 1 if inverse else 0, _ptr(e), _ptr(out), self._sp())
        return DevArray(out)

    def _get_obs(self, exit_='D', row_mask=None):
        """n_env == 1: the reference's call — with multi_display the caller has put ego_dynamics / all_vehicles /
        v_light (already in the ego's frame, multi_ego.py:94-96) on the object and exit_ only renames the routes.
        A batch with exit_ other than 'D' is the 12-ego scene in one call: `_ego`, the candidates and the light are
        WORLD values; every env's ego goes into its exit's frame (kept as `_ego_exit`), the candidates follow
        (rotation in float64, route renaming, light rule of multi_ego.py:89-92) inside eb_get_obs."""
        cand, cmode = self._cand, self._cand_mode
        ego, exit_ids = self._ego, None
        if self.n_env == 1 and (self.multi_display or exit_ != 'D'):
            inj = self._absorb_injected(exit_)
            if inj[0] is not None:
                cand, cmode = inj
        elif self.n_env > 1 and not (isinstance(exit_, str) and exit_ == 'D'):
            exit_ids = self._exit_ids(exit_)
            self._ego_exit = torch.empty_like(self._ego)
            self.api.exit_frame(self._h, self.n_env, _ptr(exit_ids), 0, _ptr(self._ego), _ptr(self._ego_exit), self._sp())
            ego = self._ego_exit
        m = cand.shape[1]
        ri = self._ref_idx
        if self.n_env == 1:
            ri = torch.tensor([int(self.ref_path.ref_index)], dtype=torch.int32, device=self.device)
        # row_mask (a masked reset): only those rows are recomputed, the others keep the observation they have
        self.api.get_obs(self._h, self.n_env, _ptr(ego), _ptr(ri), 0, m, _ptr(cand.contiguous()),
                         _ptr(cmode.contiguous()), _ptr(self._v_light), _ptr(self._virtual), _ptr(exit_ids), _ptr(row_mask),
                         _ptr(self._obs), self._sp())
        return self._ret(self._obs.clone())

    # -- step (E2E:132-144) -----------------------------------------------------------------------
    def _action_transformation_for_end2end(self, action):  # E2E:258-267
        act = _dev(np.asarray(action, np.float32).reshape(self.n_env, 2) if not isinstance(action, (torch.Tensor, DevArray))
                   else action, self.device).reshape(self.n_env, 2).contiguous()
        out = torch.empty_like(act)
        self.api.action_transform(self._h, self.n_env, _ptr(act), _ptr(out), self._sp())
        return out

    def compute_reward(self, obs, action):  # E2E:501-507: reward only, plus the dict of 16 terms
        obs_t = _dev(obs, self.device).reshape(self.n_env, self.obs_dim).contiguous()
        act_t = _dev(action, self.device).reshape(self.n_env, 2).contiguous()
        reward, _, _, _, _, reward_dict = self.env_model.compute_rewards(obs_t, act_t)
        if self.n_env == 1:
            return reward.numpy()[0], {k: v.numpy()[0] for k, v in reward_dict.items()}
        return reward, reward_dict

    def _get_next_ego_state(self, trans_action):  # E2E:269-283
        act = _dev(trans_action, self.device).reshape(self.n_env, 2).contiguous()
        nxt, par = torch.empty_like(self._ego), torch.empty_like(self._params)
        self.api.env_ego_step(self._h, self.n_env, _ptr(self._ego), _ptr(act), _ptr(nxt), _ptr(par), self._sp())
        return nxt, par

    def _traffic_step(self):
        """SUMO's role (TRF:220-238): advance every candidate by the model's prediction step (in place), then the pool's
        re-entry rule — two kernels."""
        if not self._cand.is_contiguous():
            self._cand = self._cand.contiguous()
        flat = self._cand.reshape(self.n_env, 4 * self.n_cand)
        self.api.veh_predict(self._traffic.h, self.n_env, _ptr(flat), _ptr(flat), self._sp())
        if self.respawn and self._flows is None:
            self._respawn_counter += 1
            self.api.traffic_respawn(self._traffic.h, self.n_env, self.n_cand, _ptr(self._cand), _ptr(self._entry5),
                                     C.c_float(CROSSROAD_SIZE / 2 + 40.), C.c_float(self.POOL_EDGE_SPAN), C.c_float(EXPECTED_V),
                                     C.c_uint64(self._respawn_seed), C.c_uint64(self._respawn_counter), None, None, None,
                                     C.c_float(0.0), self._sp())

    def _judge_done(self):  # E2E:200-256 -> (done_type, done)
        code = torch.empty((self.n_env,), dtype=torch.uint8, device=self.device)
        lw = self._flows.cand_lw() if self._flows is not None else None
        self.api.judge_done(self._h, self.n_env, _ptr(self._ego), _ptr(self._params), _ptr(self._obs), self.n_cand,
                            _ptr(self._cand.contiguous()), _ptr(self._cand_mode), _ptr(lw), _ptr(self._v_light), _ptr(code),
                            self._sp())
        self.done_code = code
        if self.n_env == 1:
            c = int(code[0].item())
            return _capi.DONE_NAMES[c], int(c != 0)
        # a batch keeps the uint8 codes on the device (EB_DONE_*; done_names() spells them out on request)
        return DevArray(code), DevArray((code != 0).to(torch.uint8))

    # the single predicates of E2E:223-256 on the published host state (n_env == 1; multi_ego.py:128 calls the last one)
    def _deviate_too_much(self):
        return bool(abs(float(np.asarray(self.obs)[self.ego_info_dim])) > 15)

    def _break_road_constrain(self):
        from .endtoend_env_utils import judge_feasible
        return not all(judge_feasible(x, y, self.training_task) for x, y in self.ego_dynamics['Corner_point'])

    def _break_stability(self):
        r_bound = self.ego_dynamics['r_bound']
        return not (-r_bound < self.ego_dynamics['r'] < r_bound)

    def _break_red_light(self):
        return bool(self.v_light != 0 and self.ego_dynamics['y'] > -CROSSROAD_SIZE / 2 and self.training_task != 'right')

    def _is_achieve_goal(self):
        x, y = self.ego_dynamics['x'], self.ego_dynamics['y']
        if self.training_task == 'left':
            return bool(x < -CROSSROAD_SIZE / 2 - 10 and 0 < y < LANE_NUMBER * LANE_WIDTH)
        if self.training_task == 'right':
            return bool(x > CROSSROAD_SIZE / 2 + 10 and -LANE_NUMBER * LANE_WIDTH < y < 0)
        return bool(y > CROSSROAD_SIZE / 2 + 10 and 0 < x < LANE_NUMBER * LANE_WIDTH)

    def done_names(self):
        """done_type strings of the last step for every env of a batch (E2E:208-221)."""
        return [_capi.DONE_NAMES[int(c)] for c in self.done_code.cpu().numpy()]

    def _settle_snapshots(self):
        """before a call that writes `_ref_idx` in place: a handed-out info['ref_index'] that is still alive gets its own copy now"""
        w = self._ri_snapshot
        if w is not None:
            d = w()
            if d is not None:
                d.t
            self._ri_snapshot = None

    def _ref_index_out(self):
        """info['ref_index'] of a batch (E2E:143): copy_outputs — a snapshot made on demand (an auto-reset step and every reset()
        rewrite `_ref_idx` in place and settle it first); otherwise the array itself"""
        if not self.copy_outputs:
            return DevArray(self._ref_idx)
        w = self._ri_snapshot
        d = w() if w is not None else None
        if d is None:       # (while nothing has rewritten `_ref_idx` the steps share one object: its value is theirs)
            d = _SnapshotOnDemand(self._ref_idx)
            self._ri_snapshot = weakref.ref(d)
        return d

    def _new_step_set(self):
        B, dev = self.n_env, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        bufs = dict(act=torch.empty((B, 2), **f32), out5=torch.empty((5, B), **f32),
                    obs=torch.empty((B, self.obs_dim), **f32), code=torch.empty((B,), dtype=torch.uint8, device=dev))
        bufs['rew'] = bufs['out5'][0]                                    # the reward row as a view of its own (made once per set)
        bufs['d16'] = torch.empty((16, B), **f32) if (self._want_d16 or B == 1) else None
        bufs['final'] = torch.empty((B, self.obs_dim), **f32) if self._auto_rule is not None else None
        return bufs

    def _step_buffers(self):
        """copy_outputs: a fresh set of output arrays per step (the caching allocator hands the blocks back in microseconds; nothing
        that was handed out is ever written again).  Otherwise two pre-allocated sets used in turn: what step() hands out stays
        valid until the step after the next one, and the step loop itself allocates nothing on the device."""
        if self._ri1 is None:
            self._ri1 = torch.zeros((1,), dtype=torch.int32, device=self.device)
        if self.copy_outputs:
            return self._new_step_set()
        if self._bufs is None:
            self._bufs = [self._new_step_set() for _ in range(2)]
        self._buf_i ^= 1
        bufs = self._bufs[self._buf_i]
        if bufs['d16'] is None and self._want_d16:
            bufs['d16'] = torch.empty((16, self.n_env), dtype=torch.float32, device=self.device)
        return bufs

    def step(self, action):
        """E2E:132-144 through ONE C-ABI call and — when the candidate tile fits the LDS — ONE kernel launch (eb_env_step,
        csrc/eb_env_step.hip): action scaling -> reward on the current obs -> ego step -> traffic step -> observation ->
        done code -> the traffic pool re-enters the vehicles that left the map (the observation saw the pool as this step
        left it, the way the reference sees SUMO's state of the step).  What is handed out: arrays of their own (copy_outputs, the
        default) or two pre-allocated buffer sets used in turn (valid until the step after next) — see __init__."""
        B, dev = self.n_env, self.device
        if self._ri_snapshot is not None and self._auto_rule is not None:     # (this launch rewrites `_ref_idx`)
            self._settle_snapshots()
        raw = _unwrap_action(action, B, dev)
        bufs = self._step_buffers()
        act, out5, d16, obs_out, code, final = bufs['act'], bufs['out5'], bufs['d16'], bufs['obs'], bufs['code'], bufs['final']
        if obs_out.data_ptr() == self._obs.data_ptr():          # (after a reset wrote into the same buffer)
            self._obs = self._obs.clone()
        obs_in = self._obs
        ri = self._ref_idx
        if B == 1:
            self._ri1.fill_(int(self.ref_path.ref_index))
            ri = self._ri1
        if not self._cand.is_contiguous():
            self._cand = self._cand.contiguous()
        lw = self._flows.cand_lw() if self._flows is not None else None      # the vTypes' (l, w): TRF:263-295
        sp = self._sp()
        rs = None
        if self._flows is None and self.respawn:     # vehicles that left the map re-enter on their lane, counter-based draws
            self._respawn_counter += 1
            rs = self._respawn_rule
            rs.seed, rs.counter = self._respawn_seed, self._respawn_counter
        ar = self._auto_rule
        if ar is not None and self._flows is None:   # the envs this step finishes start their next episode in the same launch: the draws reset(mask=done) would make
            self._reset_counter += 2
            ar.seed, ar.counter = self._respawn_seed ^ self._RESET_SALT, self._reset_counter - 1
            ar.pool.seed, ar.pool.counter = self._respawn_seed ^ self._POOL_SALT, self._reset_counter
            ar.final_obs = final.data_ptr()
        elif ar is not None:                         # ... over the flow source: eb_env_reset's and eb_traffic_flow_reset's draws of that reset
            from .traffic import RESET_SALT
            self._reset_counter += 1
            self._flows.reset_counter += 1
            ar.seed, ar.counter = self._respawn_seed ^ self._RESET_SALT, self._reset_counter
            ar.flow_seed, ar.flow_counter = self._flows.seed ^ RESET_SALT, self._flows.reset_counter
            ar.v_light = self._v_light.data_ptr()
            ar.final_obs = final.data_ptr()
        fr = None
        if self._flows is not None:    # exits, emissions, the new mode bytes and the light for the NEXT step ride on the way out of the
            self._flows.cand = self._cand      # same launch (eb_flow_rule: the observation saw this step's state)
            if self.flow_in_step:
                fr = self._flows.step_rule()
        self.api.env_step(self._h, self._traffic.h, B, _ptr(obs_in), _ptr(raw), _ptr(ri), 0, _ptr(self._ego),
                          _ptr(self._params), self.n_cand, _ptr(self._cand), _ptr(self._cand_mode), _ptr(lw),
                          _ptr(self._v_light), _ptr(self._virtual), _ptr(act), _ptr(out5), _ptr(d16), _ptr(obs_out),
                          _ptr(code), C.byref(rs) if rs is not None else None, C.byref(ar) if ar is not None else None,
                          C.byref(fr) if fr is not None else None,
                          C.byref(self._time_limit) if self._time_limit is not None else None, sp)
        self._obs, self.done_code = obs_out, code
        if self._flows is not None:
            if not self.flow_in_step:          # (the same rule as a launch of its own: eb_traffic_flow_step)
                self._flows.after_step(self.api, self._traffic.h, sp)
            self._cand, self._cand_mode, self._v_light = self._flows.cand, self._flows.mode(), self._flows.v_light()
        self._publish_state()                                                           # E2E:136, 139
        keys = EnvironmentModel.REWARD_KEYS
        if B == 1:
            self.action = act[0].cpu().numpy()
            self.obs = obs_out[0].cpu().numpy()
            reward = out5[0].cpu().numpy()[0]
            d16h = d16[:, 0].cpu().numpy()
            self.reward_info = {k: d16h[i] for i, k in enumerate(keys)}                # E2E:505-507
            self.reward_info.update({'final_rew': reward})                              # E2E:142
            c = int(code[0].item())
            self.done_type, done = _capi.DONE_NAMES[c], int(c != 0)                     # E2E:141
        else:
            self.action, self.obs, reward = DevArray(act), DevArray(obs_out), DevArray(bufs['rew'])
            self.reward_info = _RewardInfo(keys, d16, reward, None if d16 is not None else
                                           (lambda: self._reward_terms(obs_in, act)))   # rows of d16, wrapped on access
            self.done_type, done = DevArray(code), _LazyDone(code)                      # 0 / 1 per env, computed when read
        all_info = {'all_vehicles': self.all_vehicles, 'ego_dynamics': self.ego_dynamics, 'v_light': self.v_light,
                    'reward_info': self.reward_info,
                    'ref_index': self.ref_path.ref_index if B == 1 else
                    self._ref_index_out()}   # E2E:143
        if self._time_limit is not None:     # gym's TimeLimit: info['TimeLimit.truncated'] = not done (E: the limit ended the episode)
            all_info['TimeLimit.truncated'] = (c == _TIME_LIMIT_CODE) if B == 1 else _LazyTruncated(code)
        if ar is not None:
            all_info['final_observation'] = DevArray(final)      # the terminal rows of the envs with done != 0
            self._injected = False
        elif self._auto_compose:
            # auto_reset over the flow source: the masked reset's launches behind the step's (env_reset, the flow source's
            # init_traffic, the masked observation) — same contract as the pool's one-launch form: the observation handed out holds
            # the reset rows, info['final_observation'] the step's own rows (terminal for the finished envs), `done` / done_type
            # / reward_info stay the step's
            kept = (self.action, self.reward_info, self.done_type)
            obs_ret = self.reset(mask=done)                      # (`done` not read yet: the step's done codes serve as the mask)
            self.action, self.reward_info, self.done_type = kept
            self.done_code = code                                # (reset() left its own array, the finished envs' codes cleared)
            all_info['final_observation'] = DevArray(obs_out)
            all_info['ref_index'] = self._ref_index_out()
            return obs_ret, reward, done, all_info
        return self.obs, reward, done, all_info

    def _reward_terms(self, obs_in, act):
        """The 16 reward terms of a step whose dict was not requested from the kernel (first read): eb_compute_rewards on the
        observation and the scaled action that step consumed — the same bits; from now on step() asks for them directly."""
        self._want_d16 = True
        B = self.n_env
        out5 = torch.empty((5, B), dtype=torch.float32, device=self.device)
        d16 = torch.empty((16, B), dtype=torch.float32, device=self.device)
        self.api.compute_rewards(self._h, B, _ptr(obs_in), _ptr(act), _ptr(out5), _ptr(d16), self._sp())
        return d16


def make(env_id='CrossroadEnd2end-v0', **kwargs):
    """gym.make's role for the one id the reference registers (README.md:55-59: `register(id='CrossroadEnd2end-v0',
    entry_point='endtoend:CrossroadEnd2end', max_episode_steps=200)`; caller: mpc/main.py:542-576 `gym.make('CrossroadEnd2end-v0',
    training_task=..., num_future_data=...)`): the env inside its episode step limit — here the limit is the env's own
    `max_episode_steps` (the step kernel keeps the count), not a wrapper object."""
    if env_id != 'CrossroadEnd2end-v0':
        raise ValueError("unknown env id %r (the reference registers 'CrossroadEnd2end-v0' only)" % (env_id,))
    kwargs.setdefault('max_episode_steps', MAX_EPISODE_STEPS)
    return CrossroadEnd2end(**kwargs)
