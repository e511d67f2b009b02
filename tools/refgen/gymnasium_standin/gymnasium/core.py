from .utils import seeding
class Env:
    metadata = {"render_modes": []}
    render_mode = None
    spec = None
    _np_random = None
    def step(self, action): raise NotImplementedError
    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random, seed = seeding.np_random(seed)
    def render(self): raise NotImplementedError
    def close(self): pass
    @property
    def unwrapped(self): return self
    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random, _ = seeding.np_random()
        return self._np_random
    @np_random.setter
    def np_random(self, v): self._np_random = v
class Wrapper(Env):
    def __init__(self, env): self.env = env
