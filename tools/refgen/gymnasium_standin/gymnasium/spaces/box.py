import numpy as np
from .space import Space
from .. import logger
def _is_num(v): return np.issubdtype(type(v), np.integer) or np.issubdtype(type(v), np.floating)
def _get_inf(dtype, sign):
    if np.dtype(dtype).kind == "f": return np.inf if sign == "+" else -np.inf
    if np.dtype(dtype).kind == "i": return np.iinfo(dtype).max - 2 if sign == "+" else np.iinfo(dtype).min + 2
    raise ValueError
def _broadcast(value, dtype, shape, inf_sign):
    if _is_num(value):
        value = _get_inf(dtype, inf_sign) if np.isinf(value) else value
        return np.full(shape, value, dtype=dtype)
    assert isinstance(value, np.ndarray)
    if np.any(np.isinf(value)):
        temp = value.astype(dtype); temp[np.isinf(value)] = _get_inf(dtype, inf_sign); value = temp
    return value
class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        assert dtype is not None
        self.dtype = np.dtype(dtype)
        if shape is not None: shape = tuple(int(d) for d in shape)
        elif isinstance(low, np.ndarray): shape = low.shape
        elif isinstance(high, np.ndarray): shape = high.shape
        elif _is_num(low) and _is_num(high): shape = (1,)
        else: raise ValueError("Box shape is inferred from low and high")
        _low = np.full(shape, low, dtype=float) if _is_num(low) else low
        self.bounded_below = -np.inf < _low
        _high = np.full(shape, high, dtype=float) if _is_num(high) else high
        self.bounded_above = np.inf > _high
        low = _broadcast(low, self.dtype, shape, "-"); high = _broadcast(high, self.dtype, shape, "+")
        assert low.shape == shape and high.shape == shape
        self._shape = shape
        self.low = low.astype(self.dtype); self.high = high.astype(self.dtype)
        super().__init__(self._shape, self.dtype, seed)
    def is_bounded(self, manner="both"):
        b, a = bool(np.all(self.bounded_below)), bool(np.all(self.bounded_above))
        return {"both": b and a, "below": b, "above": a}[manner]
    def sample(self, mask=None):
        high = self.high if self.dtype.kind == "f" else self.high.astype("int64") + 1
        sample = np.empty(self.shape)
        unbounded = ~self.bounded_below & ~self.bounded_above
        upp_bounded = ~self.bounded_below & self.bounded_above
        low_bounded = self.bounded_below & ~self.bounded_above
        bounded = self.bounded_below & self.bounded_above
        sample[unbounded] = self.np_random.normal(size=unbounded[unbounded].shape)
        sample[low_bounded] = self.np_random.exponential(size=low_bounded[low_bounded].shape) + self.low[low_bounded]
        sample[upp_bounded] = -self.np_random.exponential(size=upp_bounded[upp_bounded].shape) + self.high[upp_bounded]
        sample[bounded] = self.np_random.uniform(low=self.low[bounded], high=high[bounded], size=bounded[bounded].shape)
        if self.dtype.kind in ["i", "u", "b"]: sample = np.floor(sample)
        return sample.astype(self.dtype)
    def contains(self, x):
        if not isinstance(x, np.ndarray):
            logger.warn("Casting input x to numpy array.")
            try: x = np.asarray(x, dtype=self.dtype)
            except (ValueError, TypeError): return False
        return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))
