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


def write_acdc_like(root: str, train_patients: int = 8, val_patients: int = 3, height: int = 256, width: int = 256,
                    num_classes: int = 4, seed: int = 0) -> str:
    """Write a small dataset in the reference's on-disk format (contrastyou/dataloader/acdc_dataset.py:14-24):
    ``<root>/ACDC_contrast/{train,val}/{img,gt}/patientNNN_FF_SS.png`` (8-bit 'L' PNGs, gt = class index) plus
    ``acdc_info.npy`` ({group: slices in the volume}, 200 entries).  Stands in for ACDC where the dataset itself cannot be
    downloaded (tests, input-pipeline bench)."""
    import os

    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(seed)
    base = os.path.join(root, "ACDC_contrast")
    info = {}
    pid = 1
    for mode, count in (("train", train_patients), ("val", val_patients)):
        for sub in ("img", "gt"):
            os.makedirs(os.path.join(base, mode, sub), exist_ok=True)
        for _ in range(count):
            for frame in (0, 1):
                n_slices = int(rng.integers(6, 11))
                info[f"patient{pid:03d}_{frame:02d}"] = n_slices
                for s in range(n_slices):
                    yy, xx = np.mgrid[0:height, 0:width]
                    cy, cx, r = height / 2 + rng.normal(0, 6), width / 2 + rng.normal(0, 6), 30 + 4 * s
                    d = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
                    gt = np.zeros((height, width), np.uint8)
                    for c in range(1, num_classes):
                        gt[d < r * (num_classes - c) / (num_classes - 1)] = c
                    img = np.clip(40 + 45 * gt + rng.normal(0, 12, gt.shape), 0, 255).astype(np.uint8)
                    name = f"patient{pid:03d}_{frame:02d}_{s:02d}.png"
                    Image.fromarray(img, mode="L").save(os.path.join(base, mode, "img", name))
                    Image.fromarray(gt, mode="L").save(os.path.join(base, mode, "gt", name))
            pid += 1
    k = 900
    while len(info) < 200:      # the reference asserts 200 entries (100 patients x ED/ES)
        info[f"patient{k:03d}_00"] = 9
        k += 1
    np.save(os.path.join(base, "acdc_info.npy"), info, allow_pickle=True)
    return base
