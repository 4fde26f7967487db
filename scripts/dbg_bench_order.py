import sys, os, gc, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from env_build_amd.dynamics_and_models import EnvironmentModel
dev = torch.device('cuda', 0)
which = sys.argv[1]
def es(tag, n=4096):
    r = bench.env_step_bench(torch, dev, n)
    print(tag, 'env_step %d: %.1f us (wall %.1f)' % (n, r['avg_launch_us'], r['wall_us_per_step_incl_state_restores']), flush=True)
m16 = EnvironmentModel('left', num_future_data=0, mode='training', n_veh=16, device=dev)
if 's' in which: print('side', bench.side_config(torch, None, m16, 4096, 16, 11, 100, 25, 3)['avg_launch_us'])
if 'o' in which: bench.one_launch_forms(torch, m16, 4096, 16, 11, reps=20); print('one_launch_forms done')
if 'f' in which:
    m64 = EnvironmentModel('left', num_future_data=0, mode='training', n_veh=64, device=dev)
    print('f16', bench.side_config(torch, None, m64, 65536, 64, 12, 100, 25, 3, f16=True)['avg_launch_us'])
es('A', 65536)
es('after 65536')
es('again')
