"""GPU (-m gpu): the reference-generated fixtures whose edge inputs reached the HIP kernels only by transitivity until
round 3 (GPU == oracle on random inputs, oracle == fixture on the edge inputs) now go through the HIP library itself,
with the tolerances of tests/test_oracle_golden.py: G2 (f_xu known-answer rows: v_x = 0, a_x sign changes), G3 (vehicles at
controlled distances around 2.5 / 3.5 m, ego points around every road wall; out5 AND the 16-term dict, penalty masks
bit-exact), G4 (closest point + tracking error incl. the reference's own vector DAM:803-811), G5T (teacher-forced N = 32
steps), G9 (ss), G12 (exit frames).  Each check prints its own "[gpu]" line in the parity-margin table."""
import pytest

from tests import _golden_checks as CK
from tests._helpers import DeviceModel

pytestmark = pytest.mark.gpu
TASKS = ('left', 'straight', 'right')
TAG = '[gpu] '


def _make(task, **kw):
    return DeviceModel(task, **kw)


def test_g2_f_xu_on_gpu():
    CK.check_g2_f_xu(_make, TAG)


@pytest.mark.parametrize('task', TASKS)
def test_g3_compute_rewards_on_gpu(task):
    CK.check_g3_compute_rewards(_make, task, TAG)


def test_g4_reference_own_vector_on_gpu():
    CK.check_g4_reference_own_vector(_make, TAG)


@pytest.mark.parametrize('task', TASKS)
def test_g4_tracking_on_gpu(task):
    CK.check_g4_tracking(_make, task, TAG)


@pytest.mark.parametrize('task', TASKS)
def test_g5t_teacher_forced_n32_on_gpu(task):
    CK.check_g5t_teacher_forced_n32(_make, task, TAG)


@pytest.mark.parametrize('task', TASKS)
def test_g9_ss_on_gpu(task):
    CK.check_g9_ss(_make, task, TAG)


def test_g12_exit_frames_on_gpu():
    CK.check_g12_exit_frames(_make, TAG)


def test_g7_config1_through_the_one_launch_env_step():
    """BASELINE.json configs[0] (single env, 8 vehicles, 200 steps) through the one-launch kernel, state carried on the device side
    of the test from step to step"""
    CK.check_g7_through_env_step(_make, TAG)
