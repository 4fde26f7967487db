"""SUMO-free traffic source for the batched env: what the reference gets from SUMO over
TraCI (traffic.py:37-238 with sumo_files/cross.rou.xml and a.net.xml), restated as device-side bookkeeping over a
fixed pool of vehicle slots per env.

What is reproduced (file:line of the reference's data / behaviour):
  * the twelve flows of cross.rou.xml:18-44 — one per route, emitting a vehicle every 3600 / vehsPerHour seconds
    (SUMO's `vehsPerHour` spacing), `departPos="random"` along the 75 m approach lane (a.net.xml:97-101),
    `departSpeed="random"` in [0, maxSpeed] of the flow's vType (cross.rou.xml:2-13: 8, 8, 7 m/s), vType length /
    width (4.75 x 1.60, 4.17 x 1.78, 4.17 x 1.78);
  * the traffic light: phase pinned in 'training' mode (0, or 2 with probability 1/2 for task 'right',
    traffic.py:158-161, 222-223), otherwise the static 25 / 5 / 25 / 5 s programme of a.net.xml:145-150;
  * `init_traffic`'s removal of the vehicles in conflict with the ego's start pose (traffic.py:168-192): same two
    box tests in ego and vehicle coordinates;
  * vehicles are dropped when they leave the map and only ACTIVE slots are visible to the observation and the
    collision test (their mode id is EB_VMODE_EMPTY otherwise).
  * free-flow acceleration of the vTypes (accel = 2.6 m/s^2 up to maxSpeed, cross.rou.xml:2-13): a vehicle that
    departs slowly speeds up as a Krauss vehicle without a leader does.
What is not: SUMO's car-following, lane changing and light compliance — between emission and exit a vehicle moves
with the analytic model's own prediction step (EnvironmentModel.veh_predict, DAM:394-427: straight at its speed,
arc inside the junction), which is what the safety shield assumes about it anyway.  Vehicles are dropped 40 m past
the junction, beyond every range filter of the observation (E2E:393-411).

The per-step rule and the reset are one kernel each (eb_traffic_flow_step, eb_traffic_flow_reset; include/envbuild.h;
identical in the CPU oracle); this module holds the tables and the slot state — no arithmetic."""
import ctypes as C
import torch

from . import _capi
from .endtoend_env_utils import CROSSROAD_SIZE, LANE_NUMBER, LANE_WIDTH

# route -> (vehsPerHour, vType index): cross.rou.xml:18-44 with the edge pairs of cross.rou.xml:46-60
FLOWS = dict(dl=(800, 0), du=(800, 1), dr=(800, 2), rd=(600, 0), rl=(600, 1), ru=(800, 2),
             ur=(800, 0), ud=(800, 1), ul=(800, 2), lu=(600, 0), lr=(800, 1), ld=(800, 2))
VTYPES = ((4.754264, 1.596668, 8.0), (4.173896, 1.77515, 8.0), (4.173896, 1.77515, 7.0))   # length, width, maxSpeed
ROUTES = tuple(_capi.VMODES)                    # the twelve modes, in EB_VMODE_* order
LANE_START = 100.0                              # approach lanes run from |coord| = 100 to the stop line at 25
EXIT_RANGE = CROSSROAD_SIZE / 2 + 40.0          # a departing vehicle is dropped beyond this
ACCEL = 2.6                                     # vType accel, cross.rou.xml:2-13
LIGHT_PROGRAMME = ((25.0, 0), (5.0, 1), (25.0, 2), (5.0, 3))   # a.net.xml:145-150
RESET_SALT = 0x5DEECE66D                        # resets draw from a different stream than the per-step emissions


def approach_lane(mode):
    """(x, y, phi) at the start of the lane a vehicle of `mode` arrives on, and its unit direction: lane index by
    destination, as SUMO's departLane="best" sorts them (left turn innermost)."""
    lane = {'l': 0.5, 'u': 1.5, 'r': LANE_NUMBER - 0.5}
    s, e = mode[0], mode[1]
    if s == 'd':
        return (LANE_WIDTH * lane[e], -LANE_START, 90.), (0., 1.)
    if s == 'u':
        off = {'r': 0.5, 'd': 1.5, 'l': LANE_NUMBER - 0.5}[e]
        return (-LANE_WIDTH * off, LANE_START, -90.), (0., -1.)
    if s == 'r':
        off = {'d': 0.5, 'l': 1.5, 'u': LANE_NUMBER - 0.5}[e]
        return (LANE_START, LANE_WIDTH * off, 180.), (-1., 0.)
    off = {'u': 0.5, 'r': 1.5, 'd': LANE_NUMBER - 0.5}[e]
    return (-LANE_START, -LANE_WIDTH * off, 0.), (1., 0.)


def light_phase(sim_time):
    """Phase index of the static programme at `sim_time` seconds (tensor or float)."""
    cycle = sum(d for d, _ in LIGHT_PROGRAMME)
    t = torch.as_tensor(sim_time, dtype=torch.float64) % cycle
    out = torch.zeros_like(t, dtype=torch.uint8)
    edge = 0.0
    for d, ph in LIGHT_PROGRAMME:
        out = torch.where((t >= edge) & (t < edge + d), torch.full_like(out, ph), out)
        edge += d
    return out


class FlowTraffic(object):
    """Slots: `per_route` per route, route-major — slot j serves route ROUTES[j // per_route] for good, so the
    prediction handle's per-slot turn table is static while `mode` [B, M] switches between the route's id and
    EB_VMODE_EMPTY as vehicles come and go."""

    def __init__(self, n_env, device, generator, training_task, mode='training', per_route=5, step_time=0.1):
        self.B, self.K, self.dev, self.gen = int(n_env), int(per_route), device, generator
        self.task, self.env_mode, self.dt = training_task, mode, float(step_time)
        self.M = len(ROUTES) * self.K
        if self.M > 64:
            raise ValueError('per_route * 12 must be <= 64')
        self.slot_modes = [r for r in ROUTES for _ in range(self.K)]
        dev = device
        self.route_id = torch.tensor([_capi.VMODE_ID[m] for m in self.slot_modes], dtype=torch.uint8, device=dev)
        starts = [approach_lane(m) for m in self.slot_modes]
        self.start = torch.tensor([s[0] for s in starts], dtype=torch.float32, device=dev)          # [M, 3]
        self.dirn = torch.tensor([s[1] for s in starts], dtype=torch.float32, device=dev)           # [M, 2]
        self.period = torch.tensor([3600.0 / FLOWS[r][0] for r in ROUTES], dtype=torch.float32, device=dev)   # [12]
        vt = [VTYPES[FLOWS[m][1]] for m in self.slot_modes]
        self.lw = torch.tensor([[v[0], v[1]] for v in vt], dtype=torch.float32, device=dev)          # [M, 2]
        self.vmax = torch.tensor([v[2] for v in vt], dtype=torch.float32, device=dev)               # [M]
        self.lane5 = torch.cat([self.start, self.dirn], 1).contiguous()                              # [M, 5]
        self.cand = torch.zeros((self.B, self.M, 4), dtype=torch.float32, device=dev)
        self.active = torch.zeros((self.B, self.M), dtype=torch.uint8, device=dev)
        self.timer = torch.zeros((self.B, len(ROUTES)), dtype=torch.float32, device=dev)
        self.sim_step = torch.zeros((self.B,), dtype=torch.int32, device=dev)
        self.phase0 = torch.zeros((self.B,), dtype=torch.uint8, device=dev)
        self.emitted = torch.zeros((self.B, len(ROUTES)), dtype=torch.int32, device=dev)
        self._mode = torch.full((self.B, self.M), _capi.VMODE_EMPTY, dtype=torch.uint8, device=dev)
        self._vlight = torch.zeros((self.B,), dtype=torch.uint8, device=dev)
        self.veh_len = self.lw[:, 0].contiguous()                                                   # [M]
        self._lw_b = None
        self.seed, self.counter, self.reset_counter = 0x5EED, 0, 0
        self._rule = None

    # -- views the env hands to the kernels -----------------------------------------------------------
    def mode(self):
        return self._mode

    def cand_lw(self):
        """(l, w) of every slot's vType for the collision test (TRF:263-295 reads veh['l'], veh['w']): [B, M, 2]"""
        if self._lw_b is None:
            self._lw_b = self.lw.expand(self.B, self.M, 2).contiguous()
        return self._lw_b

    def v_light(self):
        return self._vlight

    @property
    def sim_time(self):
        return self.sim_step.to(torch.float64) * self.dt

    # -- dynamics: both rules are ONE kernel each (include/envbuild.h); nothing is computed here ---------
    def reset(self, api, handle, rows, ego, stream):
        """(Re)start the traffic of the envs whose byte in `rows` (uint8 [B], None = all) is set, around the ego start
        poses `ego` [B, 6]: eb_traffic_flow_reset — presence draws, random depart position / speed, init_traffic's
        conflict removal (TRF:168-192), timers, clock and light."""
        self.reset_counter += 1
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        api.traffic_flow_reset(handle, self.B, self.K, p(rows), p(ego), p(self.cand), p(self.active), p(self.timer),
                               p(self.emitted), p(self.sim_step), p(self.phase0), p(self.lane5), p(self.period), p(self.vmax),
                               p(self.veh_len), C.c_float(LANE_START - CROSSROAD_SIZE / 2), 1 if self.task == 'right' else 0,
                               1 if self.env_mode == 'training' else 0, C.c_uint64(self.seed ^ RESET_SALT),
                               C.c_uint64(self.reset_counter), p(self._mode), p(self._vlight), stream)

    def step_rule(self):
        """The same bookkeeping as the last stage of the env's own step launch: struct eb_flow_rule for eb_env_step (ABI 4) — the
        arrays are this object's, the counter advances per step."""
        self.counter += 1
        if self._rule is None:
            p = lambda t: t.data_ptr()
            self._rule = _capi.EbFlowRule(self.K, p(self.active), p(self.timer), p(self.emitted), p(self.sim_step), p(self.lane5),
                                          p(self.period), p(self.vmax), self.dt, EXIT_RANGE, ACCEL, LANE_START - CROSSROAD_SIZE / 2,
                                          0 if self.env_mode == 'training' else 1, 0, 0, p(self._mode), p(self._vlight))
        self._rule.seed, self._rule.counter = self.seed, self.counter
        return self._rule

    def after_step(self, api, handle, stream):
        """Bookkeeping after the env has advanced every slot by one prediction step — exits, free-flow acceleration,
        emissions, clock and light — as ONE kernel (eb_traffic_flow_step)."""
        self.counter += 1
        p = lambda t: C.c_void_p(t.data_ptr())
        api.traffic_flow_step(handle, self.B, self.K, p(self.cand), p(self.active), p(self.timer), p(self.emitted),
                              p(self.sim_step), p(self.lane5), p(self.period), p(self.vmax), C.c_float(self.dt),
                              C.c_float(EXIT_RANGE), C.c_float(ACCEL), C.c_float(LANE_START - CROSSROAD_SIZE / 2),
                              0 if self.env_mode == 'training' else 1, C.c_uint64(self.seed), C.c_uint64(self.counter),
                              p(self._mode), p(self._vlight), stream)
