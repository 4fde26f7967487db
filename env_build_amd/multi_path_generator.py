"""hierarchical_decision/multi_path_generator.py:23-39 of the reference: the three candidate ReferencePath
objects of a task, one per exit lane, each with its path selected (the static generator the drivers use; the
file's older dynamic Bezier planners are dead code in the reference and are not reproduced)."""
from .dynamics_and_models import ReferencePath


class MultiPathGenerator(object):
    def __init__(self, ref_index=3, device=None):
        self.path_num = 3                                   # number of trajectories
        self.exp_v = 8.
        self.order = [0 for _ in range(self.path_num)]
        self.ego_info_dim = 6
        self.ref_index = ref_index
        self.path_list = []
        self._device = device

    def generate_path(self, task):
        self.path_list = []
        for path_index in range(self.path_num):
            ref = ReferencePath(task, device=self._device)
            ref.set_path(path_index)
            self.path_list.append(ref)
        return self.path_list
