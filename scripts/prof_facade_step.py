#!/usr/bin/env python3
"""Profiling aid: where the Python side of CrossroadEnd2end.step / reset(mask=done) spends its time (cProfile, n_env = 4096)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from env_build_amd.endtoend import CrossroadEnd2end
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = CrossroadEnd2end('left', n_env=B, multi_display=True)
env.reset()
act = torch.rand((B, 2), device=env.device) * 2 - 1
def loop(n):
    for _ in range(n):
        obs, r, done, info = env.step(act)
        env.reset(mask=done)
loop(200)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); loop(3000); pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(22)
