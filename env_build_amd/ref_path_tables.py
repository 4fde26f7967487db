"""Host-side construction of the reference-path tables (ReferencePath._construct_ref_path,
DAM:598-700): per task three candidate paths (one start lane x three end lanes), each a 1199-point
straight approach + a cubic Bezier through the junction + a 1199-point straight exit, sampled at
30 points per metre, with heading = atan2 of forward differences in degrees.

Runs once per ReferencePath in NumPy (the reference does the same); the tables are then uploaded
with eb_set_paths and live in device memory / LDS.  The Bezier is evaluated in float64 with the
barycentric Horner scheme published by the `bezier` package the reference depends on (unpinned,
not vendored), then cast to fp32 exactly where the reference casts (DAM:619).
"""
from math import pi

import numpy as np

from .endtoend_env_utils import CROSSROAD_SIZE, LANE_NUMBER, LANE_WIDTH

SL = 40                    # straight length [m], DAM:599
METER_POINTNUM_RATIO = 30  # DAM:600


def cubic_bezier(nodes, s_vals):
    """nodes [2,4] (fp32 control points, DAM:613-615) -> [2, len(s_vals)] float64."""
    nodes = np.asarray(nodes, dtype=np.float64)
    s = np.asarray(s_vals, dtype=np.float64)
    lam1, lam2 = 1.0 - s, s
    res = np.outer(nodes[:, 0], lam1)
    res = (res + np.outer(nodes[:, 1], 3.0 * lam2)) * lam1
    lam2_sq = lam2 * lam2
    res = (res + np.outer(nodes[:, 2], 3.0 * lam2_sq)) * lam1
    return res + np.outer(nodes[:, 3], lam2 * lam2_sq)


def _assemble(start_x, start_y, mid, end_x, end_y):
    xs = np.append(np.append(start_x, mid[0]), end_x)   # DAM:624-625
    ys = np.append(np.append(start_y, mid[1]), end_y)
    xs_1, ys_1 = xs[:-1], ys[:-1]
    xs_2, ys_2 = xs[1:], ys[1:]
    phis_1 = np.arctan2(ys_2 - ys_1, xs_2 - xs_1) * 180 / pi   # fp32, DAM:629-630
    return xs_1, ys_1, phis_1.astype(np.float32)


def build_ref_paths(task):
    """-> (path_list [3 x (xs, ys, phis) fp32], path_len_list, control_points)."""
    half = CROSSROAD_SIZE / 2
    n_sl = SL * METER_POINTNUM_RATIO
    ones = np.ones(shape=(n_sl,), dtype=np.float32)
    path_list, path_len_list, control_points = [], [], []
    if task == 'left':                                        # DAM:602-633
        control_ext = CROSSROAD_SIZE / 3.
        start_offset = LANE_WIDTH * 0.5
        for end_offset in [LANE_WIDTH * (i + 0.5) for i in range(LANE_NUMBER)]:
            cps = [(start_offset, -half), (start_offset, -half + control_ext),
                   (-half + control_ext, end_offset), (-half, end_offset)]
            n_mid = int(pi / 2 * (half + LANE_WIDTH / 2)) * METER_POINTNUM_RATIO
            start_x = LANE_WIDTH / 2 * ones[:-1]
            start_y = np.linspace(-half - SL, -half, n_sl, dtype=np.float32)[:-1]
            end_x = np.linspace(-half, -half - SL, n_sl, dtype=np.float32)[1:]
            end_y = end_offset * ones[1:]
            control_points.append(cps)
            path_list.append((cps, n_mid, start_x, start_y, end_x, end_y))
    elif task == 'straight':                                  # DAM:635-665
        control_ext = CROSSROAD_SIZE / 3.
        start_offset = LANE_WIDTH * 1.5
        for end_offset in [LANE_WIDTH * (i + 0.5) for i in range(LANE_NUMBER)]:
            cps = [(start_offset, -half), (start_offset, -half + control_ext),
                   (end_offset, half - control_ext), (end_offset, half)]
            n_mid = CROSSROAD_SIZE * METER_POINTNUM_RATIO
            start_x = start_offset * ones[:-1]
            start_y = np.linspace(-half - SL, -half, n_sl, dtype=np.float32)[:-1]
            end_x = end_offset * ones[1:]
            end_y = np.linspace(half, half + SL, n_sl, dtype=np.float32)[1:]
            control_points.append(cps)
            path_list.append((cps, n_mid, start_x, start_y, end_x, end_y))
    else:                                                     # DAM:667-700
        if task != 'right':
            raise ValueError("task must be 'left', 'straight' or 'right', got %r" % (task,))
        control_ext = CROSSROAD_SIZE / 5.
        start_offset = LANE_WIDTH * (LANE_NUMBER - 0.5)
        for end_offset in [-LANE_WIDTH * 2.5, -LANE_WIDTH * 1.5, -LANE_WIDTH * 0.5]:
            cps = [(start_offset, -half), (start_offset, -half + control_ext),
                   (half - control_ext, end_offset), (half, end_offset)]
            n_mid = int(pi / 2 * (half - LANE_WIDTH * (LANE_NUMBER - 0.5))) * METER_POINTNUM_RATIO
            start_x = start_offset * ones[:-1]
            start_y = np.linspace(-half - SL, -half, n_sl, dtype=np.float32)[:-1]
            end_x = np.linspace(half, half + SL, n_sl, dtype=np.float32)[1:]
            end_y = end_offset * ones[1:]
            control_points.append(cps)
            path_list.append((cps, n_mid, start_x, start_y, end_x, end_y))

    out = []
    for cps, n_mid, start_x, start_y, end_x, end_y in path_list:
        node = np.asfortranarray([[p[0] for p in cps], [p[1] for p in cps]], dtype=np.float32)
        mid = cubic_bezier(node, np.linspace(0, 1.0, n_mid)).astype(np.float32)   # DAM:616-619
        xs, ys, phis = _assemble(start_x.astype(np.float32), start_y, mid, end_x.astype(np.float32)
                                 if end_x.dtype != np.float32 else end_x, end_y)
        out.append((np.ascontiguousarray(xs, dtype=np.float32),
                    np.ascontiguousarray(ys, dtype=np.float32),
                    np.ascontiguousarray(phis, dtype=np.float32)))
        path_len_list.append((n_sl, n_mid, len(xs)))                              # DAM:633
    return out, path_len_list, control_points
