import numpy as np
from collections.abc import Sequence
from .space import Space
class Tuple(Space, Sequence):
    def __init__(self, spaces, seed=None):
        self.spaces = tuple(spaces)
        super().__init__(None, None, seed)
    def seed(self, seed=None):
        seeds = []
        if isinstance(seed, Sequence) and not isinstance(seed, (str, bytes)):
            for s, sp in zip(seed, self.spaces): seeds += sp.seed(s)
        elif isinstance(seed, int):
            seeds = super().seed(seed)
            subseeds = self.np_random.integers(np.iinfo(np.int32).max, size=len(self.spaces))
            for sp, ss in zip(self.spaces, subseeds): seeds.extend(sp.seed(int(ss)))
        elif seed is None:
            for sp in self.spaces: seeds.extend(sp.seed(seed))
        else: raise TypeError
        return seeds
    def sample(self, mask=None): return tuple(s.sample() for s in self.spaces)
    def contains(self, x):
        if isinstance(x, (list, np.ndarray)): x = tuple(x)
        return isinstance(x, tuple) and len(x) == len(self.spaces) and all(s.contains(p) for s, p in zip(self.spaces, x))
    def __getitem__(self, i): return self.spaces[i]
    def __len__(self): return len(self.spaces)
