import numpy as np


def np_random(seed=None):
    """gymnasium>=0.26 semantics: PCG64 Generator seeded through a SeedSequence."""
    seed_seq = np.random.SeedSequence(seed)
    np_seed = seed_seq.entropy
    rng = np.random.Generator(np.random.PCG64(seed_seq))
    return rng, np_seed
