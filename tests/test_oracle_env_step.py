"""CPU (-m "not gpu"): the oracle's eb_env_step composite (incl. the re-entry rule and the nullable outputs of ABI 3)
equals its own single calls — the same check the GPU suite runs on the one-launch kernel (tests/_env_step_check.py)."""
import pytest

from tests._env_step_check import CASES, composite_case, masked_obs_case, reset_pool_case, respawn_conflict_case, wrap_guard_case
from tests._helpers import HostModel


@pytest.mark.parametrize('task,B,M,NV,nf', CASES[3:7])
def test_oracle_env_step_composite(oracle, task, B, M, NV, nf):
    composite_case(lambda t, **kw: HostModel(oracle, t, **kw), task, B, M, NV, nf)


@pytest.mark.parametrize('task', ['left', 'straight', 'right'])
def test_oracle_masked_observation_pass(oracle, task):
    masked_obs_case(lambda t, **kw: HostModel(oracle, t, **kw), task)


def test_oracle_pool_reset_keeps_clear_of_the_ego(oracle):
    respawn_conflict_case(lambda t, **kw: HostModel(oracle, t, **kw))


@pytest.mark.parametrize('task', ['left', 'straight', 'right'])
def test_oracle_reset_pool_composite(oracle, task):
    reset_pool_case(lambda t, **kw: HostModel(oracle, t, **kw), task)


@pytest.mark.timeout(60)
def test_angle_wrap_loops_are_bounded(oracle):
    wrap_guard_case(HostModel(oracle, 'left'))
