#!/usr/bin/env python3
"""Debug aid: per-step time of CrossroadEnd2end.step next to scene statistics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from env_build_amd.endtoend import CrossroadEnd2end
B = 65536
env = CrossroadEnd2end('left', n_env=B, multi_display=True)
env.reset()
act = torch.rand((B, 2), device=env.device) * 2 - 1
for t in range(45):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.step(act)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e6
    e = env._ego
    x, y, v = e[:, 3], e[:, 4], e[:, 0]
    c = env._cand
    print('step %2d %6.1f us | ego x [%.0f, %.0f] y [%.0f, %.0f] v mean %.1f nan %d | done!=0 %.2f | cand |x|<60&|y|<60 %.2f | light %.2f virt %.2f' % (
        t, dt, x.min(), x.max(), y.min(), y.max(), v.mean(), int(torch.isnan(e).any(1).sum()), (env.done_code != 0).float().mean(),
        ((c[..., 0].abs() < 60) & (c[..., 1].abs() < 60)).float().mean(), (env._v_light != 0).float().mean(), (env._virtual != 0).float().mean()))
