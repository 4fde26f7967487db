"""CPU: host-side pieces of the callers around the hot path — env_build_amd/recorder.py against fixture G10 (the
reference's own Recorder.record on the same inputs, oracle/gen_golden_recorder.py) and the on-disk layout the
reference's tools read (utils/recorder.py:93-108); the flow / light tables of env_build_amd/traffic.py; the path
hysteresis rule of env_build_amd/hier_decision.py."""
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


def test_flow_tables_light_programme_and_lanes():
    """Host tables of env_build_amd/traffic.py against sumo_files/cross.rou.xml / a.net.xml values (restated here)."""
    from env_build_amd.traffic import FLOWS, LIGHT_PROGRAMME, ROUTES, VTYPES, approach_lane, light_phase
    assert len(ROUTES) == 12 and set(FLOWS) == set(ROUTES)
    assert sorted(v[0] for v in FLOWS.values()).count(600) == 3 and sorted(v[0] for v in FLOWS.values()).count(800) == 9   # cross.rou.xml:18-44
    assert {FLOWS[r][0] for r in ('rd', 'rl', 'lu')} == {600}
    assert [v[2] for v in VTYPES] == [8.0, 8.0, 7.0] and abs(VTYPES[0][0] - 4.754264) < 1e-9
    assert LIGHT_PROGRAMME == ((25.0, 0), (5.0, 1), (25.0, 2), (5.0, 3))           # a.net.xml:145-150
    assert [int(light_phase(t)) for t in (0, 24.9, 25.0, 29.9, 30.0, 54.9, 55.0, 59.9, 60.0, 85.0)] == [0, 0, 1, 1, 2, 2, 3, 3, 0, 1]
    # lanes of a.net.xml:97-101: 1o_2 (x = 1.88) is the left-turn lane, 1o_0 (x = 9.38) the right-turn lane
    (x, y, phi), d = approach_lane('dl')
    assert abs(x - 1.875) < 1e-6 and y == -100.0 and phi == 90.0 and d == (0., 1.)
    assert abs(approach_lane('dr')[0][0] - 9.375) < 1e-6 and abs(approach_lane('du')[0][0] - 5.625) < 1e-6
    for m in ROUTES:     # every lane starts 100 m out and points at the junction
        (x, y, phi), (dx, dy) = approach_lane(m)
        assert max(abs(x), abs(y)) == 100.0 and x * dx + y * dy < 0


def test_path_hysteresis_rule_on_cpu_tensors():
    """HierarchicalDecision.select_path (hier_decision.py:118-121 per env): keep the old path unless the best one is
    better by at least 0.1 — the method is plain tensor arithmetic, checked here against the reference's scalar rule."""
    import torch
    from types import SimpleNamespace
    from env_build_amd.hier_decision import HierarchicalDecision
    rng = np.random.default_rng(2)
    B = 500
    pv = rng.uniform(0, 1, (3, B)).astype(np.float32)
    pv[:, :50] = pv[0, :50]                                   # ties: argmin takes the first, nothing beats the old path
    pv[1, 50:100] = pv[0, 50:100] - np.float32(0.1)           # exactly on the threshold
    old = rng.integers(0, 3, B)
    fake = SimpleNamespace(old_index=torch.from_numpy(old))
    got = HierarchicalDecision.select_path(fake, torch.from_numpy(pv)).numpy()
    for b in range(B):
        path_values = pv[:, b]
        old_value = path_values[old[b]]
        new_index, new_value = int(np.argmin(path_values)), min(path_values)          # hier_decision.py:119
        want = old[b] if old_value - new_value < 0.1 else new_index                   # hier_decision.py:120
        assert got[b] == want, b
