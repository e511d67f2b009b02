import numpy as np
from .space import Space
class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64, seed=None, start=None):
        self.nvec = np.array(nvec, dtype=dtype, copy=True)
        self.start = np.zeros(self.nvec.shape, dtype=dtype) if start is None else np.array(start, dtype=dtype)
        super().__init__(self.nvec.shape, dtype, seed)
    def sample(self, mask=None):
        return (self.np_random.random(self.nvec.shape) * self.nvec).astype(self.dtype) + self.start
    def contains(self, x):
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.all(self.start <= x) and np.all(x - self.start < self.nvec))
