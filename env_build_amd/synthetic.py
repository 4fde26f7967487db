"""Seeded synthetic inputs for the rollout path (SURVEY.md §8(d) 'Synthetic inputs').

Host-side NumPy only; shared by bench.py, the parity tests and oracle/gen_golden.py so that the
GPU path, the CPU oracle and the reference see identical bytes.
"""
import numpy as np

from .endtoend_env_utils import VEHICLE_MODE_LIST, tiled_mode_list
from .ref_path_tables import build_ref_paths


def make_rollout_inputs(task, n_env, n_veh, horizon, seed=0, n_future=0, path_tables=None,
                        near_fraction=0.25):
    """-> dict(ego [B,6], veh [B,4N], ref_idx [B] int32, actions [H,B,2], modes [N] str).

    Per env: ref_index ~ U{0,1,2}; path index ~ U{700..2099} (mirrors E2E:474); ego pose = path
    point + N(0, 0.3 m) on x,y and N(0, 3 deg) on phi; v_x ~ U(0,8) (E2E:482), v_y = r = 0.
    Vehicles: `near_fraction` of the slots within 12 m of the ego (exercises the penalty
    branches), the rest uniform in [-60,60]^2; v ~ U(0,8); phi ~ U(-180,180].  The tracking
    columns are NOT produced here — callers fill them with tracking_error_vector.
    """
    rng = np.random.default_rng(seed)
    paths = path_tables if path_tables is not None else build_ref_paths(task)[0]
    B, N = int(n_env), int(n_veh)
    ref_idx = rng.integers(0, 3, size=B).astype(np.int32)
    max_index = min(len(p[0]) for p in paths) - 1
    pidx = np.minimum(rng.integers(700, 2100, size=B), max_index)
    assert len({len(p[0]) for p in paths}) == 1   # the three paths of a task share one length
    px = np.stack([p[0] for p in paths])[ref_idx, pidx]
    py = np.stack([p[1] for p in paths])[ref_idx, pidx]
    pphi = np.stack([p[2] for p in paths])[ref_idx, pidx]
    ego = np.zeros((B, 6), np.float32)
    ego[:, 0] = rng.uniform(0, 8, B)
    ego[:, 3] = px + rng.normal(0, 0.3, B)
    ego[:, 4] = py + rng.normal(0, 0.3, B)
    ego[:, 5] = pphi + rng.normal(0, 3.0, B)

    veh = np.zeros((B, N, 4), np.float32)
    near = rng.random((B, N)) < near_fraction
    rad = 12.0 * np.sqrt(rng.random((B, N)))
    ang = rng.uniform(-np.pi, np.pi, (B, N))
    far_x = rng.uniform(-60, 60, (B, N))
    far_y = rng.uniform(-60, 60, (B, N))
    veh[:, :, 0] = np.where(near, ego[:, 3:4] + rad * np.cos(ang), far_x)
    veh[:, :, 1] = np.where(near, ego[:, 4:5] + rad * np.sin(ang), far_y)
    veh[:, :, 2] = rng.uniform(0, 8, (B, N))
    veh[:, :, 3] = -rng.uniform(-180, 180, (B, N))      # (-180, 180]
    actions = rng.uniform(-1, 1, (int(horizon), B, 2)).astype(np.float32)
    modes = VEHICLE_MODE_LIST[task] if N == len(VEHICLE_MODE_LIST[task]) else tiled_mode_list(task, N)
    return dict(ego=ego, veh=veh.reshape(B, 4 * N), ref_idx=ref_idx, actions=actions,
                modes=list(modes), n_future=int(n_future))


def assemble_obs(ego, tracking, veh):
    """[ego 6 | tracking 3(n+1) | veh 4N] row layout of DAM:356."""
    return np.ascontiguousarray(np.concatenate([ego, tracking, veh], axis=1), dtype=np.float32)
