"""Minimal space descriptors for the vector env (gymnasium is not a dependency here).

They mirror what the reference exposes through ``env.observation_space`` /
``env.action_space`` — spaces/discrete_extended.py, spaces/box_extended.py,
spaces/image_multi_discrete.py — each with its own numpy generator seeded at construction.
"""
import numpy as np


def _gen(seed):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))


class DiscreteSpace:
    def __init__(self, n, seed=None):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)
        self.np_random = _gen(seed)

    def sample(self, max=None, prob=None, size=1, replace=True):
        """DiscreteExtended.sample (spaces/discrete_extended.py:11-23)."""
        if max is None:
            max = self.n
        s = np.squeeze(self.np_random.choice(max, size=size, p=prob, replace=replace))
        return int(s) if s.shape == () else s

    def contains(self, x):
        return bool(np.issubdtype(np.asarray(x).dtype, np.integer) and 0 <= int(x) < self.n)

    def __repr__(self):
        return f"DiscreteSpace({self.n})"


class TupleSpace:
    """TupleExtended (spaces/tuple_extended.py): seeding it re-seeds every sub-space the way
    gymnasium's Tuple.seed(int) does (third-party, 0.29 / 1.x): one int32 sub-seed per sub-space
    drawn from Generator(seed)."""

    def __init__(self, spaces, seed=None):
        self.spaces = tuple(spaces)
        self.shape = None
        self.dtype = None
        self.np_random = _gen(seed)
        if isinstance(seed, int):
            subseeds = self.np_random.integers(np.iinfo(np.int32).max, size=len(self.spaces))
            for sp, ss in zip(self.spaces, subseeds):
                sp.np_random = _gen(int(ss))

    def sample(self):
        return tuple(sp.sample() for sp in self.spaces)

    def contains(self, x):
        return len(x) == len(self.spaces) and all(sp.contains(v) for sp, v in zip(self.spaces, x))

    def __getitem__(self, i):
        return self.spaces[i]

    def __len__(self):
        return len(self.spaces)

    def __repr__(self):
        return "TupleSpace(" + ", ".join(repr(sp) for sp in self.spaces) + ")"


class BoxSpace:
    def __init__(self, low, high, shape, dtype=np.float32, seed=None):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)
        self.np_random = _gen(seed)

    def sample(self):
        """gymnasium Box.sample for an all-bounded or all-unbounded box."""
        if np.all(np.isfinite(self.low)) and np.all(np.isfinite(self.high)):
            s = self.np_random.uniform(low=self.low, high=self.high, size=self.shape)
        else:
            s = self.np_random.normal(size=self.shape)
        return s.astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape
                    and np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"BoxSpace({self.low.flat[0]}, {self.high.flat[0]}, {self.shape}, {self.dtype})"


class ImageSpace(BoxSpace):
    def __init__(self, width, height):
        super().__init__(0, 255, (width, height, 1), dtype=np.uint8)


class BatchedSpace:
    """The space of a batch of `n` independent copies of `single` (what gymnasium's
    vector.utils.batch_space builds for VectorEnv.observation_space / action_space): shape
    (n,) + single.shape, sample() stacks n samples of the single space, contains() checks every row."""

    def __init__(self, single, n):
        self.single, self.n = single, int(n)
        self.dtype = single.dtype
        self.shape = None if single.shape is None else (self.n,) + tuple(single.shape)

    def sample(self):
        rows = [self.single.sample() for _ in range(self.n)]
        if isinstance(self.single, TupleSpace):
            return tuple(np.asarray(col) for col in zip(*rows))
        return np.stack([np.asarray(r) for r in rows])

    def contains(self, x):
        if isinstance(self.single, TupleSpace):
            return len(x) == len(self.single) and all(
                len(col) == self.n and all(sp.contains(v) for v in col) for sp, col in zip(self.single.spaces, x))
        x = np.asarray(x)
        return x.shape[0] == self.n and all(self.single.contains(row) for row in x)

    def __repr__(self):
        return f"BatchedSpace({self.single!r}, {self.n})"
