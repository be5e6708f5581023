"""FixRandomSeed (ref whl:deepclustering2/decorator/decorator.py:196-212): seed numpy + random inside the
block, restore both states on exit."""
import random

import numpy as np


class FixRandomSeed:
    def __init__(self, random_seed: int = 0):
        self.random_seed = random_seed
        self.randombackup = random.getstate()
        self.npbackup = np.random.get_state()

    def __enter__(self):
        np.random.seed(self.random_seed)
        random.seed(self.random_seed)

    def __exit__(self, *_):
        np.random.set_state(self.npbackup)
        random.setstate(self.randombackup)
