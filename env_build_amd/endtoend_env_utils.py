"""Map / vehicle constants, mode tables and scalar geometry helpers of the crossroad scene.

Mirrors the names of the reference's endtoend_env_utils.py (UTL:14-46, 73-157, 232-237) so that
callers can switch imports; the SUMO coordinate conversions (UTL:199-229) are out of scope
(SURVEY.md §2 row 5).
"""
import math
from collections import OrderedDict

L, W = 4.8, 2.0                      # UTL:14
LANE_WIDTH = 3.75                    # UTL:15
LANE_NUMBER = 3                      # UTL:16
CROSSROAD_SIZE = 50                  # UTL:17
EXPECTED_V = 8.                      # UTL:18

# slots per vehicle mode in the observation, in observation order (UTL:21-23)
VEHICLE_MODE_DICT = dict(left=OrderedDict(dl=2, du=2, ud=2, ul=2),
                         straight=OrderedDict(dl=1, du=2, ud=2, ru=2, ur=2),
                         right=OrderedDict(dr=1, ur=2, lr=2))


def dict2flat(inp):
    out = []
    for key, val in inp.items():
        out.extend([key] * val)
    return out


def dict2num(inp):
    return sum(inp.values())


VEH_NUM = {task: dict2num(d) for task, d in VEHICLE_MODE_DICT.items()}            # UTL:40-42
VEHICLE_MODE_LIST = {task: dict2flat(d) for task, d in VEHICLE_MODE_DICT.items()}  # UTL:44-46

ROUTE2MODE = {('1o', '2i'): 'dr', ('1o', '3i'): 'du', ('1o', '4i'): 'dl',
              ('2o', '1i'): 'rd', ('2o', '3i'): 'ru', ('2o', '4i'): 'rl',
              ('3o', '1i'): 'ud', ('3o', '2i'): 'ur', ('3o', '4i'): 'ul',
              ('4o', '1i'): 'ld', ('4o', '2i'): 'lr', ('4o', '3i'): 'lu'}         # UTL:56-59
MODE2ROUTE = {m: r for r, m in ROUTE2MODE.items()}                                 # UTL:68-71
TASK2ROUTEID = {'left': 'dl', 'straight': 'du', 'right': 'dr'}                     # UTL:66
MODE2TASK = {'dr': 'right', 'du': 'straight', 'dl': 'left',
             'rd': 'left', 'ru': 'right', 'rl': ' straight',
             'ud': 'straight', 'ur': 'left', 'ul': 'right',
             'ld': 'right', 'lr': 'straight', 'lu': 'left'}                        # UTL:61-64


def tiled_mode_list(task, n_veh):
    """Slot modes for a non-native slot count: VEHICLE_MODE_LIST[task] tiled to n_veh
    (SURVEY.md §7 'N_veh != 8/9/5'; the reference itself only knows the native list)."""
    base = VEHICLE_MODE_LIST[task]
    return [base[i % len(base)] for i in range(n_veh)]


def judge_feasible(orig_x, orig_y, task):  # UTL:73-104
    half = CROSSROAD_SIZE / 2
    in_middle = -half < orig_y < half and -half < orig_x < half
    if task == 'left':
        return bool((0 < orig_x < LANE_WIDTH and orig_y <= -half)
                    or (0 < orig_y < LANE_WIDTH * LANE_NUMBER and orig_x < -half) or in_middle)
    elif task == 'straight':
        return bool((LANE_WIDTH < orig_x < LANE_WIDTH * 2 and orig_y <= -half)
                    or (0 < orig_x < LANE_WIDTH * LANE_NUMBER and orig_y >= half) or in_middle)
    else:
        assert task == 'right'
        return bool((LANE_WIDTH * 2 < orig_x < LANE_WIDTH * 3 and orig_y <= -half)
                    or (-LANE_WIDTH * LANE_NUMBER < orig_y < 0 and orig_x > half) or in_middle)


def shift_coordination(orig_x, orig_y, coordi_shift_x, coordi_shift_y):  # UTL:107-117
    return orig_x - coordi_shift_x, orig_y - coordi_shift_y


def deal_with_phi(phi):  # UTL:232-237
    while phi > 180:
        phi -= 360
    while phi <= -180:
        phi += 360
    return phi


def rotate_coordination(orig_x, orig_y, orig_d, coordi_rotate_d):  # UTL:120-142
    rad = coordi_rotate_d * math.pi / 180
    transformed_x = orig_x * math.cos(rad) + orig_y * math.sin(rad)
    transformed_y = -orig_x * math.sin(rad) + orig_y * math.cos(rad)
    return transformed_x, transformed_y, deal_with_phi(orig_d - coordi_rotate_d)


def shift_and_rotate_coordination(orig_x, orig_y, orig_d, coordi_shift_x, coordi_shift_y,
                                  coordi_rotate_d):  # UTL:145-149
    sx, sy = shift_coordination(orig_x, orig_y, coordi_shift_x, coordi_shift_y)
    return rotate_coordination(sx, sy, orig_d, coordi_rotate_d)


def rotate_and_shift_coordination(orig_x, orig_y, orig_d, coordi_shift_x, coordi_shift_y,
                                  coordi_rotate_d):  # UTL:152-157
    rx, ry, d = rotate_coordination(orig_x, orig_y, orig_d, coordi_rotate_d)
    tx, ty = shift_coordination(rx, ry, coordi_shift_x, coordi_shift_y)
    return tx, ty, d


# ---- frames of the other exits (multi_ego.py:84-96 observes each ego in its own exit-relative frame) ----
def transform_vehicles(vehicles, x, y, rotate_d):  # UTL:160-180 (cal_info_in_transform_coordination)
    """Every vehicle dict re-expressed in the frame shifted by (x, y) and then rotated by rotate_d degrees
    (positive = anticlockwise); speed, size and route are carried over."""
    out = []
    for veh in vehicles:
        sx, sy = shift_coordination(veh['x'], veh['y'], x, y)
        tx, ty, tphi = rotate_coordination(sx, sy, veh['phi'], rotate_d)
        out.append(dict(veh, x=tx, y=ty, phi=tphi))
    return out


def transform_ego(ego_dynamics, x, y, rotate_d):  # UTL:183-196 (cal_ego_info_in_transform_coordination)
    """The ego's pose and corner points in the shifted-then-rotated frame; updates and returns the dict, as
    the reference does."""
    heading = ego_dynamics['phi']
    sx, sy = shift_coordination(ego_dynamics['x'], ego_dynamics['y'], x, y)
    tx, ty, tphi = rotate_coordination(sx, sy, heading, rotate_d)
    corners = []
    for cx, cy in ego_dynamics['Corner_point']:
        csx, csy = shift_coordination(cx, cy, x, y)
        ctx, cty, _ = rotate_coordination(csx, csy, heading, rotate_d)
        corners.append((ctx, cty))
    ego_dynamics.update(dict(x=tx, y=ty, phi=tphi, Corner_point=corners))
    return ego_dynamics


# the reference's names
cal_info_in_transform_coordination = transform_vehicles
cal_ego_info_in_transform_coordination = transform_ego
