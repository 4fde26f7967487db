#!/usr/bin/env python3
"""Debugging aid: repeat the env-step configuration whose `params` output differed once in a while, and say how it differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from env_build_amd import _capi
from env_build_amd.endtoend_env_utils import VEHICLE_MODE_LIST
from tests._helpers import DeviceModel
from tests._env_step_check import random_scene
task, B, M, NV, nf, tile = 'straight', 200, 33, 16, 0, 1
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
native = VEHICLE_MODE_LIST[task]
modes = [native[i % len(native)] for i in range(M)]
ego, cand, _, _, light, _, ref = random_scene(task, B, M, 44)
cmode = np.tile(np.array([_capi.VMODE_ID[m] for m in modes], np.uint8), (B, 1))
rng = np.random.default_rng(2)
raw = rng.uniform(-1.2, 1.2, (B, 2)).astype(np.float32)
m, tr = DeviceModel(task, mode='training', n_future=nf, n_veh=NV), DeviceModel(task, n_veh=M, modes=modes)
m.set_tile(tile)
obs0 = m.get_obs(ego, cand, cmode, light, ref_idx=ref)
act = m.action_transform(raw)
ego1, par1 = m.env_ego_step(ego, act)
ego2, par2 = m.env_ego_step(ego1, act)          # what the parameters would be if computed from the NEW ego
entry = rng.uniform(-60, 60, (M, 5)).astype(np.float32)
rule = dict(entry=entry, limit=65.0, span=60.0, v_max=8.0, seed=0x1234567, counter=9)
bad = 0
for r in range(reps):
    for kw in (dict(), dict(respawn=rule, want_scaled=False, want_dict=False)):
        got = m.env_step(tr, obs0, raw, ego, cand, cmode, ref_idx=ref, **kw)
        par, eg = got[4], got[3]
        if not np.array_equal(par, par1) or not np.array_equal(eg, ego1):
            bad += 1
            rows = np.where((par != par1).any(1) | (eg != ego1).any(1))[0]
            print('rep %d %s: rows %s' % (r, 'got7' if kw else 'got', rows[:12]), 'params equal those of the NEW ego on these rows:',
                  np.array_equal(par[rows], par2[rows]), 'ego ok:', np.array_equal(eg, ego1), flush=True)
            print('   got', par[rows[0]], 'want', par1[rows[0]], 'from-new-ego', par2[rows[0]])
print('%d repetitions x 2 calls, %d mismatching calls' % (reps, bad))
