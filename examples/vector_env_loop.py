"""vector_env_loop.py — the reference's Gym env as a batch: what a trainer that used `CrossroadEnd2end` one env at a time
(E2E:44-144) writes against env_build_amd to step thousands of them per launch.

    env = CrossroadEnd2end('left', n_env=4096, auto_reset=True)    # same constructor, two more arguments
    obs = env.reset()                                     # DevArray [B, D] (device memory; .numpy() / .t for host / torch views)
    obs, reward, done, info = env.step(actions)           # ONE kernel launch (csrc/eb_env_step.hip): the step AND the reset of the
                                                          # envs it finished — obs holds their reset observation,
                                                          # info['final_observation'] their terminal one (vector-env convention)

Without auto_reset the driver resets by hand, `obs = env.reset(mask=done)` — a second launch (`done` may be handed over as it is:
its done codes serve as the mask).  By default every value handed out is an array of its own (keep it in a replay buffer as long
as you like); `copy_outputs=False` switches to two pre-allocated output sets used in turn — zero allocations per step, values
valid until the step after next.  Everything below the Python calls runs in libenvbuild_hip.so; there is no CPU path.
Run: python examples/vector_env_loop.py [n_env] [steps] [manual]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch                                               # noqa: E402
from env_build_amd.endtoend import CrossroadEnd2end        # noqa: E402


def run(n_env=4096, steps=200, task='left', seed=0, policy=None, auto_reset=True, copy_outputs=False):
    """-> dict(episodes, mean_return, steps_per_s).  `policy(obs_tensor) -> actions [B, 2] in [-1, 1]` (default: random).
    (copy_outputs=False: this loop consumes every value before the next step.)"""
    env = CrossroadEnd2end(task, n_env=n_env, auto_reset=auto_reset, copy_outputs=copy_outputs)
    env.seed(seed)
    env.reset()                                            # (as in the reference, a reset observation is built with the flags of the
    obs = env.reset()                                      # episode before, E2E:116-126: the second reset after seed() is reproducible)
    g = torch.Generator(device=env.device).manual_seed(seed)
    ret = torch.zeros(n_env, device=env.device)           # running return per env
    finished_ret, episodes = torch.zeros((), device=env.device), torch.zeros((), device=env.device)
    t0 = None
    for k in range(-10, steps):                            # ten untimed iterations first: torch loads its kernels on first use
        if k == 0:
            ret.zero_(); finished_ret.zero_(); episodes.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        act = policy(obs.t) if policy is not None else torch.rand((n_env, 2), device=env.device, generator=g) * 2 - 1
        obs, reward, done, info = env.step(act)
        ret += reward.t
        fin = env.done_code != 0                           # the uint8 done codes of this step (env.done_names() spells them)
        finished_ret += (ret * fin).sum()
        episodes += fin.sum()
        ret = torch.where(fin, torch.zeros_like(ret), ret)
        if not auto_reset:
            obs = env.reset(mask=done)                     # the finished envs start a new episode, the others are untouched
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = float(episodes)
    return dict(episodes=int(n), mean_return=float(finished_ret) / max(n, 1.0), steps_per_s=n_env * steps / dt)


if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    print(run(B, K, auto_reset=not (len(sys.argv) > 3 and sys.argv[3] == 'manual')))
