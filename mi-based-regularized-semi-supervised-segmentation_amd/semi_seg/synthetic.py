"""Synthetic ACDC-shaped loaders: batches packed exactly as the reference loaders pack them
(``[[img, tgt], [img2, tgt2]], filenames, partitions, groups`` -- contrastyou/dataloader/acdc_dataset.py:26-32),
so the epochers run unchanged without the dataset.  Used by bench.py, the smoke test and ``main.py Data.name=synthetic``."""
from __future__ import annotations

import torch


class SyntheticPairs:
    """Infinite iterator of twice-transformed batches; tensors are generated once per length and reused
    (pinned host memory), like an infinite sampler over a small dataset."""

    def __init__(self, batch_size: int, size: int = 256, num_classes: int = 4, seed: int = 0, pool: int = 4, device=None):
        g = torch.Generator().manual_seed(seed)
        self._items = []
        for k in range(pool):
            img = torch.rand(batch_size, 1, size, size, generator=g)
            tgt = torch.randint(0, num_classes, (batch_size, 1, size, size), generator=g)
            if device is not None:
                img, tgt = img.to(device), tgt.to(device)
            elif torch.cuda.is_available():
                img, tgt = img.pin_memory(), tgt.pin_memory()
            names = [f"patient{(seed * 131 + k * batch_size + i) % 100:03d}_00_{i:02d}" for i in range(batch_size)]
            groups = [n[:13] for n in names]
            self._items.append([[[img, tgt], [img, tgt]], names, ["0"] * batch_size, groups])
        self._i = 0

    def __iter__(self):
        return self

    def __next__(self):
        item = self._items[self._i % len(self._items)]
        self._i += 1
        return item


class SyntheticEval:
    """Finite loader of single-transformed 'patients' for the eval epochers."""

    def __init__(self, num_patients: int = 2, slices: int = 4, size: int = 256, num_classes: int = 4, seed: int = 1):
        g = torch.Generator().manual_seed(seed)
        self._batches = []
        for p in range(num_patients):
            img = torch.rand(slices, 1, size, size, generator=g)
            tgt = torch.randint(0, num_classes, (slices, 1, size, size), generator=g)
            names = [f"patient{p:03d}_00_{i:02d}" for i in range(slices)]
            self._batches.append([[img, tgt], names, ["0"] * slices, [f"patient{p:03d}_00"] * slices])

    def __len__(self):
        return len(self._batches)

    def __iter__(self):
        return iter(self._batches)
