# PROBE ONLY: minimal stand-in for gymnasium (0.29-style semantics), written from memory of the public API.
from . import error, logger
from .utils import seeding
from . import utils, spaces
from .core import Env, Wrapper
from . import envs, wrappers
def register_envs(*a, **k): pass
def make(*a, **k): raise error.DependencyNotInstalled("gymnasium stand-in: make() unavailable")
