"""Batched form of the reference's model-predictive safety shield — `is_safe` / `safe_shield` of
hierarchical_decision/hier_decision.py:89-107 (5 steps, veh2veh4real) and multi_env/multi_ego.py:187-209 (20 steps,
real_punish_term): roll the analytic model forward under the caller's policy and call a start state unsafe when the
chosen penalty turns positive anywhere on the way.  The reference does this for ONE observation per call; here every
row of `obses` is checked at once, one rollout-kernel launch per step with the policy call in between (closed loop).

The policy is the caller's: any callable obs [B, D] (DevArray / torch tensor on the model's device) -> actions [B, 2]
in [-1, 1] (the reference's `policy.run_batch`).  Host glue only — the arithmetic is EnvironmentModel.rollout_out."""
import torch

from .dynamics_and_models import DevArray, _unwrap

PENALTIES = {'veh2veh4real': 4, 'real_punish_term': 3}       # index into rollout_out's 6-tuple (DAM:126)
SAFE_ACTION = (0., -1.)                                        # action_safe_set, hier_decision.py:100


def is_safe(model, policy, obses, path_index=None, steps=5, penalty='veh2veh4real'):
    """-> (safe [B] bool DevArray, accumulated penalty [B] DevArray).  `path_index` selects the path for a model in
    'selecting' mode (model.add_traj, hier_decision.py:91); a 'training'-mode model keeps its ref_indexes."""
    if penalty not in PENALTIES:
        raise ValueError('penalty must be one of %s' % sorted(PENALTIES))
    if path_index is not None:
        model.add_traj(obses, path_index)
    else:
        model.reset(obses, model.ref_indexes)
    punish = None
    for _ in range(int(steps)):
        out = model.rollout_out(policy(model.obses))
        p = _unwrap(out[PENALTIES[penalty]])
        punish = p.clone() if punish is None else punish + p
    return DevArray(~(punish > 0)), DevArray(punish)


def safe_shield(model, policy, obses, path_index=None, steps=5, penalty='veh2veh4real'):
    """-> (actions [B, 2] DevArray, shield_started [B] bool DevArray): the policy's action where the look-ahead is
    safe, the fallback action (0, -1) elsewhere (hier_decision.py:99-107)."""
    obs0 = model._obs(obses, model.state_dtype)                 # on the model's device, whatever the caller handed in
    safe, _ = is_safe(model, policy, obs0, path_index, steps, penalty)
    act = _unwrap(policy(DevArray(obs0))).to(device=obs0.device, dtype=torch.float32)
    fallback = torch.tensor(SAFE_ACTION, dtype=torch.float32, device=act.device).expand_as(act)
    started = ~_unwrap(safe)
    return DevArray(torch.where(started.unsqueeze(1), fallback, act)), DevArray(started)
