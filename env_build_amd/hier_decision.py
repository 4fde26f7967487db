"""Batched counterpart of the reference's decision loop — `HierarchicalDecision` of
hierarchical_decision/hier_decision.py:30-135 — the caller the hot path exists for: every step, for each of the
three candidate paths build the observation (`env._get_obs` after `set_traj`, :113-115), value them with the
policy's obj_v network (:116-117), keep the old path unless another is better by 0.1 (:118-121), run the
model-predictive safety shield on the chosen observation (:99-107, 126-127) and step the env with the safe
action (:131).  Here the B envs of a CrossroadEnd2end batch take those decisions together:

    3 x eb_get_obs (one per path)  ->  eb_mlp_forward on [3B, D] (obj_v)  ->  per-env hysteresis argmin  ->
    eb_shield_is_safe (5 x [policy MLP -> rollout step])  ->  eb_policy_run_batch  ->  eb_env_step

Per-env path choice means the model tracks `ref_indexes[i]` per row — EnvironmentModel's 'training' mode
semantics (DAM:342-353) with the chosen index, which for one env is exactly the reference's `add_traj(obs, i)`.
Rendering (:62-228) and video export are out of scope; the Recorder keeps the reference's log layout."""
import time

import numpy as np
import torch

from .dynamics_and_models import DevArray, EnvironmentModel, _ptr, _unwrap
from .endtoend import CrossroadEnd2end
from .multi_path_generator import MultiPathGenerator
from .policy import LoadPolicy
from .recorder import Recorder
from .shield import SAFE_ACTION, is_safe

__all__ = ['HierarchicalDecision']


class HierarchicalDecision(object):
    def __init__(self, task, train_exp_dir=None, ite=None, logdir=None, policy=None, n_env=1, device=None,
                 shield_steps=5, penalty='veh2veh4real', auto_reset=True, **env_kwargs):
        self.task = task
        self.policy = policy if policy is not None else LoadPolicy(train_exp_dir, ite, device=device)   # :33
        self.env = CrossroadEnd2end(training_task=self.task, mode='testing', n_env=n_env, device=device, **env_kwargs)
        self.device, self.n_env = self.env.device, int(n_env)
        self.model = EnvironmentModel(self.task, mode='training', n_veh=self.env.veh_num, device=self.device)   # :35
        self.recorder = Recorder(self.n_env)
        self.episode_counter = -1
        self.step_counter = -1
        self.stg = MultiPathGenerator(device=self.device)
        self.path_list = self.stg.generate_path(self.task)                              # :52
        self.logdir = logdir
        self.shield_steps, self.penalty, self.auto_reset = int(shield_steps), penalty, bool(auto_reset)
        self.step_time = self.ss_time = 0.0
        self.old_index = torch.zeros((self.n_env,), dtype=torch.int64, device=self.device)
        self.obs = None
        self.reset()

    def reset(self, mask=None):                                                         # :66-81
        self.obs = self.env.reset() if mask is None else self.env.reset(mask=mask)
        if self.logdir is not None:             # episodes are only kept when a log is being written
            self.recorder.reset(None if mask is None else _unwrap(mask).cpu().numpy())
        if mask is None:
            self.old_index.zero_()
        else:
            self.old_index.masked_fill_(_unwrap(mask).to(self.device).bool(), 0)
        if self.logdir is not None:
            self.recorder.save(self.logdir)
        self.episode_counter += 1
        return self.obs

    def is_safe(self, obs, path_index):                                                 # :89-97
        idx = _unwrap(path_index)
        if not isinstance(idx, torch.Tensor):
            idx = torch.as_tensor(np.asarray(idx).reshape(-1), device=self.device)
        self.model.reset(obs, idx.to(torch.int32))
        safe, _ = is_safe(self.model, self.policy, self.model.obses, None, self.shield_steps, self.penalty)
        return safe

    def safe_shield(self, real_obs, path_index):                                        # :99-107
        safe = _unwrap(self.is_safe(real_obs, path_index))
        act = _unwrap(self.policy.run_batch(real_obs))
        fallback = torch.tensor(SAFE_ACTION, dtype=torch.float32, device=act.device).expand_as(act)
        return DevArray(torch.where(safe.unsqueeze(1), act, fallback)), DevArray(~safe)

    def path_observations(self):
        """[3, B, D]: the observation every env would see on each candidate path (:112-115)."""
        env, B = self.env, self.n_env
        out = torch.empty((len(self.path_list), B, env.obs_dim), dtype=torch.float32, device=self.device)
        cand, cmode = env._cand.contiguous(), env._cand_mode.contiguous()
        for k in range(len(self.path_list)):
            env.api.get_obs(env._h, B, _ptr(env._ego), None, k, env.n_cand, _ptr(cand), _ptr(cmode), _ptr(env._v_light),
                            _ptr(env._virtual), None, None, _ptr(out[k]), env._sp())
        return out

    def select_path(self, path_values):
        """Hysteresis of :118-121 per env: keep the old path unless the best one is better by at least 0.1
        (values approximate minus the return, lower is better)."""
        pv = _unwrap(path_values)                                                       # [3, B]
        old_value = pv.gather(0, self.old_index.unsqueeze(0))[0]
        new_value, new_index = pv.min(0)
        return torch.where(old_value - new_value < 0.1, self.old_index, new_index)

    def step(self):                                                                     # :109-135
        self.step_counter += 1
        t0 = time.perf_counter()
        B = self.n_env
        all_obs = self.path_observations()
        path_values = _unwrap(self.policy.obj_value_batch(all_obs.reshape(-1, all_obs.shape[2]))).reshape(-1, B)
        path_index = self.select_path(path_values)
        self.old_index = path_index
        if B == 1:
            self.env.set_traj(self.path_list[int(path_index[0])])                       # :123
        self.env._ref_idx.copy_(path_index.to(torch.int32))
        self.obs_real = all_obs.gather(0, path_index.view(1, B, 1).expand(1, B, all_obs.shape[2]))[0].contiguous()
        t1 = time.perf_counter()
        safe_action, is_ss = self.safe_shield(self.obs_real, path_index)                # :126-127
        if self.logdir is not None or B == 1:
            torch.cuda.synchronize(self.device)
        self.ss_time = time.perf_counter() - t1
        self.step_time = time.perf_counter() - t0
        self.path_index, self.path_values, self.safe_action, self.is_ss = path_index, path_values, safe_action, is_ss
        if self.logdir is not None:
            self.recorder.record(self.obs_real.cpu().numpy(), safe_action.numpy(), self.step_time,
                                 path_index.cpu().numpy(), path_values.t().cpu().numpy(), self.ss_time, is_ss.numpy())   # :129-130
        env_obs, r, done, info = self.env.step(safe_action if B > 1 else safe_action.numpy()[0])   # :131
        self.obs = env_obs
        if self.auto_reset and B > 1:
            d = _unwrap(done).bool()
            if bool(d.any()):
                self.reset(mask=d)
        return done
