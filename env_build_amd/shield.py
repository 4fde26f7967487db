"""Batched form of the reference's model-predictive safety shield — `is_safe` / `safe_shield` of
hierarchical_decision/hier_decision.py:89-107 (5 steps, veh2veh4real) and multi_env/multi_ego.py:187-209 (20 steps,
real_punish_term): roll the analytic model forward under the caller's policy and call a start state unsafe when the
chosen penalty turns positive anywhere on the way.  The reference does this for ONE observation per call; here every
row of `obses` is checked at once, one rollout-kernel launch per step with the policy call in between (closed loop).

The policy is the caller's: any callable obs [B, D] (DevArray / torch tensor on the model's device) -> actions [B, 2]
in [-1, 1] (the reference's `policy.run_batch`).  When it is this package's own `LoadPolicy` (env_build_amd/policy.py:
MLPNet on the matrix cores) the whole loop is ONE C call, eb_shield_is_safe — policy kernel, rollout kernel and the
penalty accumulation enqueued back to back with no host work in between.  Host glue only — the arithmetic is
EnvironmentModel.rollout_out and the MLP kernel."""
import ctypes as C

import torch

from . import _capi
from .dynamics_and_models import DevArray, _stream, _unwrap

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
    native = _native_policy(policy)
    if native is not None and model.state_dtype == torch.float32 and native[0].device == model.device:
        return _is_safe_native(model, native, int(steps), penalty)
    punish = None
    for _ in range(int(steps)):
        out = model.rollout_out(policy(model.obses))
        p = _unwrap(out[PENALTIES[penalty]])
        punish = p.clone() if punish is None else punish + p
    return DevArray(~(punish > 0)), DevArray(punish)


def _native_policy(policy):
    """(MLPNet of the policy head, action_range) when `policy` is this package's LoadPolicy / Policy4Toyota."""
    from .policy import LoadPolicy, Policy4Toyota
    if isinstance(policy, LoadPolicy):
        policy = policy.policy
    if isinstance(policy, Policy4Toyota) and policy.policy.output_dim == 4 and getattr(policy.args, 'deterministic_policy', True):
        return policy.policy, getattr(policy.args, 'action_range', None)
    return None


def _is_safe_native(model, native, steps, penalty):
    net, action_range = native
    obs = _unwrap(model.obses)
    B = obs.shape[0]
    ri, pid = model._path_args()
    dev = model.device
    obs_a, obs_b = torch.empty_like(obs), torch.empty_like(obs)
    actions = torch.empty((B, 2), dtype=torch.float32, device=dev)
    out5 = torch.empty((5, B), dtype=torch.float32, device=dev)
    punish = torch.empty((B,), dtype=torch.float32, device=dev)
    safe = torch.empty((B,), dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    model.api.shield_is_safe(model.handle, net._handle, B, p(obs), p(ri), pid, steps, _capi.PENALTY_ID[penalty],
                             C.c_float(-1.0 if action_range is None else float(action_range)), p(obs_a), p(obs_b),
                             p(actions), p(out5), p(punish), p(safe), _stream(dev))
    model.obses = DevArray(obs_a if steps % 2 == 1 else obs_b)      # where the last step landed, as the loop leaves it
    model._after_tracking()
    return DevArray(safe.bool()), DevArray(punish)


def safe_shield(model, policy, obses, path_index=None, steps=5, penalty='veh2veh4real'):
    """-> (actions [B, 2] DevArray, shield_started [B] bool DevArray): the policy's action where the look-ahead is
    safe, the fallback action (0, -1) elsewhere (hier_decision.py:99-107)."""
    obs0 = model._obs(obses, model.state_dtype)                 # on the model's device, whatever the caller handed in
    safe, _ = is_safe(model, policy, obs0, path_index, steps, penalty)
    act = _unwrap(policy(DevArray(obs0))).to(device=obs0.device, dtype=torch.float32)
    fallback = torch.tensor(SAFE_ACTION, dtype=torch.float32, device=act.device).expand_as(act)
    started = ~_unwrap(safe)
    return DevArray(torch.where(started.unsqueeze(1), fallback, act)), DevArray(started)
