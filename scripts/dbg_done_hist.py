import sys; sys.path.insert(0,'.')
import torch, numpy as np
from env_build_amd.endtoend import CrossroadEnd2end
env = CrossroadEnd2end('left', n_env=8192, multi_display=True, traffic='pool', n_cand=16)
env.seed(0); env.reset()
g = torch.Generator(device='cpu').manual_seed(3)
for t in range(10):
    act = torch.stack([torch.rand(8192, generator=g)*0.6-0.3, torch.rand(8192, generator=g)*0.8-0.2],1).to(env.device)
    obs, r, done, info = env.step(act)
    print(t, np.bincount(env.done_code.cpu().numpy(), minlength=7))
