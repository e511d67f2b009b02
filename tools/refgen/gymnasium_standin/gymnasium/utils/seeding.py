import numpy as np
from .. import error
def np_random(seed=None):
    if seed is not None and not (isinstance(seed, int) and 0 <= seed):
        if isinstance(seed, int) is False:
            raise error.Error(f"Seed must be a python integer, actual type: {type(seed)}")
        raise error.Error(f"Seed must be greater or equal to zero, actual value: {seed}")
    seed_seq = np.random.SeedSequence(seed)
    np_seed = seed_seq.entropy
    rng = np.random.Generator(np.random.PCG64(seed_seq))
    return rng, np_seed
RNG = RandomNumberGenerator = np.random.Generator
