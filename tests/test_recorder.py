"""CPU: env_build_amd/recorder.py against fixture G10 (the reference's own Recorder.record on the same inputs,
oracle/gen_golden_recorder.py) and the on-disk layout the reference's tools read (utils/recorder.py:93-108)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from env_build_amd.recorder import Recorder  # noqa: E402
from tests._helpers import golden  # noqa: E402


def test_record_matches_the_reference_rows():
    g = golden('g10_recorder')
    assert list(g['val2record']) == Recorder.val2record
    r = Recorder()
    for t in range(len(g['obs'])):
        r.record(g['obs'][t], g['act'][t], g['cal_time'][t], g['ref_index'][t], g['path_values'][t], g['ss_time'][t], g['is_ss'][t])
    rows = r.val_list_for_an_episode
    numeric = np.array([[float(v) for j, v in enumerate(row) if j != 14] for row in rows], np.float64)
    assert np.array_equal(numeric, g['rows_numeric'])
    assert np.array_equal(np.array([row[14] for row in rows]), g['rows_path_values'])


def test_save_layout_is_what_the_reference_loader_iterates(tmp_path):
    g = golden('g10_recorder')
    B = 3
    r = Recorder(n_env=B)
    for t in range(4):
        r.record(np.stack([g['obs'][t + i] for i in range(B)]), np.stack([g['act'][t + i] for i in range(B)]), 0.01, [0, 1, 2],
                 np.stack([g['path_values'][t + i] for i in range(B)]), 0.002, [False, True, False])
        if t == 1:
            r.reset(mask=[False, True, False])            # env 1 finishes an episode early
    r.reset()
    r.save(str(tmp_path))
    a = np.load(os.path.join(str(tmp_path), 'data_across_all_episodes.npy'), allow_pickle=True)   # recorder.py:98
    assert len(a) == 4 and sorted(len(ep) for ep in a) == [2, 2, 4, 4]
    # the reference's reader: [vals_in_a_timestep[index] for vals_in_a_timestep in episode2plot] for each of the 17 names
    for ep in a:
        cols = [np.array([step[j] for step in ep]) for j in range(len(Recorder.val2record))]
        assert cols[14].shape == (len(ep), 3) and cols[0].shape == (len(ep),)
    r2 = Recorder()
    r2.load(str(tmp_path))
    tab = r2.episode_table(0)
    assert set(tab) == set(Recorder.val2record) and tab['path_values'].shape[1] == 3
