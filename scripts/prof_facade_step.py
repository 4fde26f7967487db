#!/usr/bin/env python3
"""Profiling aid: where the Python side of CrossroadEnd2end.step / reset(mask=done) spends its time (cProfile, n_env = 4096).
usage: prof_facade_step.py [n_env] [auto] [nocopy]   (auto: auto_reset=True, no reset call; nocopy: copy_outputs=False)"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from env_build_amd.endtoend import CrossroadEnd2end
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
AUTO, NOCOPY = 'auto' in sys.argv, 'nocopy' in sys.argv
env = CrossroadEnd2end('left', n_env=B, multi_display=True, auto_reset=AUTO, copy_outputs=not NOCOPY)
env.reset()
act = torch.rand((B, 2), device=env.device) * 2 - 1
def loop(n):
    for _ in range(n):
        obs, r, done, info = env.step(act)
        if not AUTO:
            env.reset(mask=done)
loop(200)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); loop(3000); pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(22)
