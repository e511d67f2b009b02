"""mdp_playground_amd — batched, MI355X-native RLToyEnv.step()/reset().

The hot path of automl/mdp-playground (mdp_playground/envs/rl_toy_env.py: step :1992,
reset :2217) as hand-written HIP kernels for gfx950 behind the reference's own
Gym/config-dict API.  See DESIGN.md.
"""
__version__ = "0.1.0"

from .mdp import build_mdp  # noqa: F401


def __getattr__(name):
    # The vector env needs torch + the HIP library; import lazily so that the host-side
    # table generator stays importable on machines without either.
    if name in ("RLToyVectorEnv", "make_vec"):
        from . import vector_env
        return getattr(vector_env, name)
    raise AttributeError(name)
