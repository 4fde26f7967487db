"""The reference's episode log — utils/recorder.py:21-99 `Recorder` — for one env or a batch: the same 17 values
per step in the same order (`val2record`), the same transformations (steer in degrees, a_x scaled, side-slip
angle beta), and the same on-disk layout: `logdir/data_across_all_episodes.npy` = an object array of episodes,
each a list of steps, each a 17-entry array (`np.load(..., allow_pickle=True)`), so the reference's plotting
tools read these logs unchanged.  With n_env > 1 every env keeps its own current episode; `reset(mask)` closes
the episodes of the envs that were reset.  Host-side bookkeeping only (NumPy); the plots themselves
(utils/recorder.py:101-290, seaborn/matplotlib) are out of scope."""
import math
import os

import numpy as np


class Recorder(object):
    val2record = ['v_x', 'v_y', 'r', 'x', 'y', 'phi', 'steer', 'a_x', 'delta_y', 'delta_v', 'delta_phi',
                  'cal_time', 'ref_index', 'beta', 'path_values', 'ss_time', 'is_ss']          # recorder.py:23-25

    def __init__(self, n_env=1):
        self.n_env = int(n_env)
        self.ego_info_dim = 6
        self.per_tracking_info_dim = 3
        self.num_future_data = 0
        self.data_across_all_episodes = []
        self._current = [[] for _ in range(self.n_env)]

    @property
    def val_list_for_an_episode(self):          # the reference's attribute (single env)
        return self._current[0]

    def reset(self, mask=None):                 # recorder.py:50-56
        rows = range(self.n_env) if mask is None else np.flatnonzero(np.asarray(mask).reshape(self.n_env))
        for i in rows:
            if self._current[i]:
                self.data_across_all_episodes.append(self._current[i])
                self._current[i] = []

    def record(self, obs, act, cal_time, ref_index, path_values, ss_time, is_ss):   # recorder.py:58-73
        obs = np.asarray(obs, np.float32).reshape(self.n_env, -1)
        act = np.asarray(act, np.float32).reshape(self.n_env, 2)
        ref_index = np.asarray(ref_index).reshape(self.n_env)
        path_values = np.asarray(path_values, np.float32)
        path_values = path_values.reshape(1, -1) if self.n_env == 1 else path_values.reshape(self.n_env, -1)
        is_ss = np.asarray(is_ss).reshape(self.n_env)
        for i in range(self.n_env):
            v_x, v_y, r, x, y, phi = obs[i, :self.ego_info_dim]
            delta_y, delta_phi, delta_v = obs[i, self.ego_info_dim:self.ego_info_dim + 3]
            steer, a_x = act[i, 0] * 0.4, act[i, 1] * 2.25 - 0.75
            beta = 0 if v_x == 0 else np.arctan(v_y / v_x) * 180 / math.pi
            steer = steer * 180 / math.pi
            row = np.empty(len(self.val2record), dtype=object)
            row[:] = [v_x, v_y, r, x, y, phi, steer, a_x, delta_y, delta_phi, delta_v, cal_time, ref_index[i], beta,
                      path_values[i], ss_time, is_ss[i]]
            self._current[i].append(row)

    def save(self, logdir):                     # recorder.py:93-95
        os.makedirs(logdir, exist_ok=True)
        out = np.empty(len(self.data_across_all_episodes), dtype=object)
        out[:] = [list(ep) for ep in self.data_across_all_episodes] if len(out) else []
        np.save(os.path.join(logdir, 'data_across_all_episodes.npy'), out, allow_pickle=True)

    def load(self, logdir):                     # recorder.py:97-99
        self.data_across_all_episodes = list(np.load(os.path.join(logdir, 'data_across_all_episodes.npy'),
                                                     allow_pickle=True))

    def episode_table(self, i):
        """Episode i as a dict name -> per-step array (what plot_and_save_ith_episode_curves builds, recorder.py:103-108)."""
        ep = self.data_across_all_episodes[i]
        return {k: np.array([step[j] for step in ep]) for j, k in enumerate(self.val2record)}
