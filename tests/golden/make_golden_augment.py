#!/usr/bin/env python
"""Golden vectors for the input pipeline by RUNNING THE REFERENCE'S OWN transform objects.

Build container only (needs /root/reference):   python tests/golden/make_golden_augment.py

The reference's ``semi_seg/augment.py`` builds ``ACDCStrongTransforms`` from its ``SequentialWrapper(Twice)``
(contrastyou/augment/sequential_wrapper.py) and the wheel's ``pil_augment`` classes; those import torchvision 0.7, which is
neither under /root/reference nor installed.  Its published thin wrappers over Pillow (Compose, ColorJitter, functional
rotate/crop/flip/to_tensor/adjust_*) are supplied from oracle/augment.py at the import seam; everything else -- the seed
plumbing, FixRandomSeed, get_params of every pil_augment transform, ToLabel, and Pillow itself -- is the real thing.
Only DATA is written: synthetic u8 slices, item seeds and the tensors the reference returned (as u8 / label bytes).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
import make_golden as MG  # noqa: E402
from oracle import augment as OA  # noqa: E402


def slices():
    out = []
    for k, (h, w) in enumerate([(256, 256), (232, 280)]):
        y, x = np.mgrid[0:h, 0:w]
        img = ((x * 3 + y * 5 + 17 * k) % 256).astype(np.uint8)
        gt = (((x // 23 + y // 19) + k) % 4).astype(np.uint8)
        out.append((img, gt))
    return out


def main():
    MG.import_reference()
    import torchvision.transforms as T  # the hollow stand-in; give it the restated 0.7 pieces
    import torchvision.transforms.functional as TF
    T.Compose, T.ColorJitter, T.ToTensor = OA.Compose, OA.ColorJitter, OA.ToTensor
    for name in ("rotate", "crop", "center_crop", "hflip", "vflip", "to_tensor"):
        setattr(TF, name, getattr(OA.tvf, name))
    from deepclustering2.augment import pil_augment
    pil_augment.tf, pil_augment.Compose = OA.tvf, OA.Compose
    from PIL import Image
    from semi_seg.augment import ACDCStrongTransforms  # the reference's own presets

    arrays = {}
    seeds = [0, 123, 99999]
    for k, (img, gt) in enumerate(slices()):
        arrays[f"img{k}"], arrays[f"gt{k}"] = img, gt
        for name in ("pretrain", "label", "val", "trainval"):
            tf = getattr(ACDCStrongTransforms, name)
            for seed in {"pretrain": seeds, "val": seeds[:1]}.get(name, seeds[1:2]):
                pi, pg = Image.fromarray(img, mode="L"), Image.fromarray(gt, mode="L")
                out = tf(imgs=[pi], targets=[pg], global_seed=seed) if name != "val" else [tf(imgs=[pi], targets=[pg])]
                for v, (ti, tg) in enumerate(out):
                    u8 = torch.round(ti * 255).to(torch.uint8)
                    assert torch.equal(ti, u8.float().div(255)) and ti.dtype == torch.float32 and tg.dtype == torch.int64
                    assert ti.shape == (1, 224, 224) and tg.shape == (1, 224, 224)
                    arrays[f"{name}/{k}/{seed}/{v}/img"] = u8[0].numpy()
                    arrays[f"{name}/{k}/{seed}/{v}/gt"] = tg[0].to(torch.uint8).numpy()
    arrays["seeds"] = np.array(seeds)
    MG.save("augment", **arrays)


if __name__ == "__main__":
    main()
