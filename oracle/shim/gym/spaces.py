import numpy as np
from collections import OrderedDict


class Box(object):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.dtype = low, high, dtype
        self.shape = shape if shape is not None else np.shape(low)

    def sample(self):
        return np.random.uniform(-1, 1, self.shape).astype(self.dtype)


class Dict(OrderedDict):
    pass
