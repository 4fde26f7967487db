#!/usr/bin/env python3
"""Profiling aid: per-wave timeline of ONE eb_rollout_step launch (wall-clock marks written by the kernel)."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from env_build_amd.dynamics_and_models import EnvironmentModel
from env_build_amd.synthetic import make_rollout_inputs

ap = argparse.ArgumentParser()
ap.add_argument('--n-env', type=int, default=65536); ap.add_argument('--n-veh', type=int, default=32)
ap.add_argument('--waves', type=int, default=5, help='waves per block of the kernel variant in use')
ap.add_argument('--tile', type=int, default=-1, help='eb_debug_set_tile: 0 = 4x8, 1 = 4x4, 2 = 1x4 (then --waves 2), -1 = by batch size')
ap.add_argument('--sched', default='-1,-1', help='eb_debug_set_rollout_sched rolling,by_progress')
ap.add_argument('--lib', default=None, help='A/B aid: bind this build of libenvbuild_hip.so instead of the in-tree one')
a = ap.parse_args()
W = a.waves
dev = torch.device('cuda', 0)
if a.lib:
    from env_build_amd import _capi
    _capi._hip_api = _capi.CApi(a.lib)
inp = make_rollout_inputs('left', a.n_env, a.n_veh, 25, seed=0)
m = EnvironmentModel('left', 0, mode='training', n_veh=a.n_veh, device=dev)
m.api.debug_set_tile(m.handle, a.tile)
m.api.debug_set_rollout_sched(m.handle, *[int(x) for x in a.sched.split(',')])
ego = torch.from_numpy(inp['ego']).to(dev); ref = torch.from_numpy(inp['ref_idx']).to(dev)
trk = m.ref_path.tracking_error_vector_batched(ego[:, 3].contiguous(), ego[:, 4].contiguous(), ego[:, 5].contiguous(),
                                               ego[:, 0].contiguous(), 0, ref_indexes=ref).t
obs0 = torch.cat([ego, trk, torch.from_numpy(inp['veh']).to(dev)], 1).contiguous()
tape = torch.from_numpy(inp['actions']).to(dev)
bufs = [torch.empty_like(obs0), torch.empty_like(obs0)]; out5 = torch.empty((25, 5, a.n_env), device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib = m.api.lib
def step(t):
    src = obs0 if t == 0 else bufs[(t - 1) & 1]
    assert lib.eb_rollout_step(m.handle, a.n_env, p(src), p(tape[t]), p(ref), 1, p(bufs[t & 1]), p(out5[t]), None, sp) == 0
for t in range(10): step(t)
torch.cuda.synchronize()
tr = torch.zeros((65536 * W // 8, 8), dtype=torch.int64, device=dev)
m.api.debug_set_trace(m.handle, p(tr), tr.numel())
step(10); step(11)          # warm the marked code path
torch.cuda.synchronize()
tr.zero_(); torch.cuda.synchronize()
# three back-to-back launches, each with its own mark buffer: the dead time between consecutive kernels
trs = [torch.zeros_like(tr) for _ in range(3)]
for k in range(3):
    m.api.debug_set_trace(m.handle, p(trs[k]), trs[k].numel()); step(12 + k)
torch.cuda.synchronize()
m.api.debug_set_trace(m.handle, None, 0)
spans = []
for x in trs:
    x = x.cpu().numpy(); used = x[x[:, 0] > 0]
    spans.append((used[:, 0].min(), used.max()))
for k in range(1, 3):
    print('launch %d: first wave start .. last wave end = %.2f us; dead time since the previous kernel\'s last wave = %.2f us'
          % (k, (spans[k][1] - spans[k][0]) / 100.0, (spans[k][0] - spans[k - 1][1]) / 100.0))
tr = trs[1]
t = tr.cpu().numpy()
nb = int((t[:, 0] > 0).sum()) // W
t = t[:nb * W].astype(np.float64)
t0 = t[:, 0].min()
entry = t[:, 6].copy()         # record waves of a build with the entry mark (EB_X & 1): when they reached their first instruction
t = (t - t0) / 100.0   # us (100 MHz)
t[t < 0] = np.nan
env = t[0::W]; rec = np.concatenate([t[i::W] for i in range(1, W)])
def q(x): return ' '.join('%6.2f' % v for v in np.nanpercentile(x, [0, 10, 50, 90, 100]))
print('blocks %d; times in us since the first wave started; percentiles 0 10 50 90 100' % nb)
for i, name in enumerate(['loads issued', 'ego published', 'bicycle step done', 'closest point found', 'head stored', 'record waves done', 'end']):
    print('env wave  %-22s %s' % (name, q(env[:, i])))
for i, name in enumerate(['loads issued', 'first record stored', 'last record stored', 'ego seen', 'near tests done', 'end']):
    print('rec wave  %-22s %s' % (name, q(rec[:, i])))
is_rec = np.arange(nb * W) % W != 0
if (entry[is_rec] > 0).all():
    en = (entry[is_rec] - entry[is_rec].min()) / 100.0
    li = (tr.cpu().numpy()[:nb * W, 0].astype(np.float64)[is_rec] - entry[is_rec].min()) / 100.0
    print('rec wave  %-22s %s   (since the first record wave entered)' % ('entered the kernel', q(en)))
    print('rec wave  %-22s %s' % ('loads issued (same zero)', q(li)))
    print('rec wave  %-22s %s' % ('entry -> loads issued', q(li - en)))
    bw = en.reshape(W - 1, nb) if False else None
blk_start = np.nanmin(t[:, 0].reshape(nb, W), axis=1)
blk_end = np.nanmax(t.reshape(nb, W * 8), axis=1)
print('block start            %s' % q(blk_start))
print('block end              %s' % q(blk_end))
print('block duration         %s' % q(blk_end - blk_start))
order = np.argsort(blk_start)
for lo, hi in ((0, nb // 4), (nb // 4, nb // 2), (nb // 2, 3 * nb // 4), (3 * nb // 4, nb)):
    sel = order[lo:hi]
    print('  blocks by start quartile: start %.2f..%.2f  mean duration %.2f  mean end %.2f  (block ids %d..%d median %d)'
          % (blk_start[sel].min(), blk_start[sel].max(), (blk_end - blk_start)[sel].mean(), blk_end[sel].mean(), sel.min(), sel.max(), int(np.median(sel))))
# placement: waves per SIMD (slot 7 = XCC_ID << 32 | HW_ID)
raw = tr.cpu().numpy()[:nb * W, 7]
hw = raw & 0xffffffff; xcc = (raw >> 32) & 0xf
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key_cu = ((xcc * 8 + se) * 2 + sh) * 16 + cu
import collections
per_simd = collections.Counter(zip(key_cu.tolist(), simd.tolist()))
per_cu = collections.Counter(key_cu.tolist())
print('CUs used: %d; waves per CU: %s' % (len(per_cu), dict(collections.Counter(per_cu.values()))))
print('waves per SIMD histogram:', dict(sorted(collections.Counter(per_simd.values()).items())), '(%d SIMDs with waves)' % len(per_simd))
envw = np.arange(nb * W) % W == 0
print('env waves per SIMD histogram:', dict(sorted(collections.Counter(collections.Counter(zip(key_cu[envw].tolist(), simd[envw].tolist())).values()).items())))
blk_simd = simd.reshape(nb, W)
print('SIMD of the waves of the first blocks:', blk_simd[:6].tolist())
