"""hierarchical_decision/multi_path_generator.py:23-39 of the reference: the candidate ReferencePath objects of a
task — one per exit lane, each with that lane's path selected.  This is the static generator the drivers use; the
file's older dynamic Bezier planners are dead code in the reference and are not reproduced."""
from .dynamics_and_models import ReferencePath
from .endtoend_env_utils import EXPECTED_V, LANE_NUMBER


class MultiPathGenerator(object):
    """Attributes as the reference's (path_num, exp_v, order, ego_info_dim, ref_index, path_list)."""

    def __init__(self, ref_index=3, device=None):
        self._device = device
        self.ref_index = ref_index
        self.path_num, self.ego_info_dim, self.exp_v = LANE_NUMBER, 6, float(EXPECTED_V)
        self.order = [0] * self.path_num
        self.path_list = []

    def _candidate(self, task, lane):
        path = ReferencePath(task, device=self._device)
        path.set_path(lane)
        return path

    def generate_path(self, task):
        self.path_list = [self._candidate(task, lane) for lane in range(self.path_num)]
        return self.path_list
