"""``FixRandomSeed(seed)``: a block in which Python's ``random`` and numpy's legacy generator start from ``seed``; both generators
continue afterwards from where they stood when the OBJECT WAS CREATED (ref whl:deepclustering2/decorator/decorator.py:196-212 takes
its snapshots in the constructor, and ``semi_seg/epocher.py:148-149,160-161,264-266`` relies on exactly that: the per-step flip
decisions are a function of the seed alone and leave the global draw sequence -- hence the next step's seed -- untouched)."""
import random

import numpy as np


class FixRandomSeed:
    def __init__(self, random_seed: int = 0):
        self.random_seed = random_seed
        self._resume = (random.getstate(), np.random.get_state())      # construction time, not __enter__ time (see above)

    def __enter__(self):
        for seed_fn in (random.seed, np.random.seed):
            seed_fn(self.random_seed)
        return self

    def __exit__(self, exc_type, exc, tb):
        py_state, np_state = self._resume
        random.setstate(py_state)
        np.random.set_state(np_state)
        return False
