import numpy as np


class Tensor(object):
    def __init__(self, arr):
        self._a = np.ascontiguousarray(arr)

    def numpy(self):
        return self._a.copy()

    def numpy_view(self):
        return self._a


def from_numpy(arr):
    return Tensor(arr)
