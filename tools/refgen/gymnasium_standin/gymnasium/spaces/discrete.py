import numpy as np
from .space import Space
class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        self.n = np.int64(n); self.start = np.int64(start)
        super().__init__((), np.int64, seed)
    def sample(self, mask=None):
        return self.start + self.np_random.integers(self.n)
    def contains(self, x):
        if isinstance(x, int): as_int64 = np.int64(x)
        elif isinstance(x, (np.generic, np.ndarray)) and (np.issubdtype(x.dtype, np.integer) and x.shape == ()): as_int64 = np.int64(x)
        else: return False
        return bool(self.start <= as_int64 < self.start + self.n)
