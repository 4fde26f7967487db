#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (container only: reads /root/reference).  Fixture G10: the reference's own
utils/recorder.py `Recorder.record` run on seeded inputs -> tests/golden/g10_recorder.npz.

The reference builds each step's row with np.array([... scalars ..., path_values, ...]) — a ragged list that the
NumPy of its day turned into an object array and today's NumPy rejects — so this script hands the module a
`np.array` that falls back to dtype=object (the legacy behaviour), and a stand-in for the absent `seaborn`
(only `sns.set` is called at import).  Nothing else of the reference is touched."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sns = types.ModuleType('seaborn')
sns.set = lambda **k: None
sys.modules.setdefault('seaborn', sns)
import matplotlib  # noqa: E402
matplotlib.use('Agg')
sys.path.insert(0, '/root/reference')
import utils.recorder as ref  # noqa: E402


class _LegacyNumpy(object):
    def __getattr__(self, k):
        return getattr(np, k)

    @staticmethod
    def array(x, *a, **k):
        try:
            return np.array(x, *a, **k)
        except ValueError:
            out = np.empty(len(x), dtype=object)
            out[:] = list(x)
            return out


ref.np = _LegacyNumpy()
rng = np.random.default_rng(10)
n = 12
obs = rng.standard_normal((n, 41)).astype(np.float32) * 5
obs[3, 0] = 0.0                                  # v_x == 0 -> beta = 0 branch
act = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
cal = rng.uniform(0, 0.05, n)
ridx = rng.integers(0, 3, n)
pv = rng.standard_normal((n, 3)).astype(np.float32)
sst = rng.uniform(0, 0.01, n)
iss = rng.integers(0, 2, n).astype(bool)
r = ref.Recorder()
r.reset()
for t in range(n):
    r.record(obs[t], act[t], cal[t], ridx[t], pv[t], sst[t], iss[t])
rows = r.val_list_for_an_episode
numeric = np.array([[float(v) for j, v in enumerate(row) if j != 14] for row in rows], np.float64)
pvals = np.array([np.asarray(row[14], np.float32) for row in rows])
np.savez(os.path.join(ROOT, 'tests', 'golden', 'g10_recorder.npz'), obs=obs, act=act, cal_time=cal, ref_index=ridx,
         path_values=pv, ss_time=sst, is_ss=iss, rows_numeric=numeric, rows_path_values=pvals,
         val2record=np.array(r.val2record))
print('g10_recorder: %d rows x %d values' % numeric.shape, r.val2record)
