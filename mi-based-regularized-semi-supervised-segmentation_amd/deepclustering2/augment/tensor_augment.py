"""TensorRandomFlip (ref whl:deepclustering2/augment/tensor_augment.py:17-45).

Decisions come from Python's ``random`` exactly as in the reference (one draw per axis per call, in
axis order) so that ``FixRandomSeed(seed)`` replays them.  ``decisions(batch)`` draws a whole batch in
the reference's order; the HIP kernels then apply them as index math (``miseg_amd.ops.flip`` or fused
into consumers) instead of a Python loop of ``clone().flip()``.
"""
import random
from typing import List

import torch


class TensorRandomFlip:
    def __init__(self, axis=None, threshold=0.5) -> None:
        if isinstance(axis, int):
            axis = [axis]
        elif axis is not None and not isinstance(axis, (list, tuple)):
            raise ValueError(str(axis))
        self._axis = axis
        assert 0 <= threshold <= 1
        self._threshold = threshold

    def decisions(self, batch: int) -> List[List[bool]]:
        """Per sample, per axis: ``random.random() < threshold`` in the reference's draw order."""
        axes = self._axis or []
        return [[random.random() < self._threshold for _ in axes] for _ in range(batch)]

    def __call__(self, tensor: torch.Tensor):
        tensor = tensor.clone()
        for ax in (self._axis or []):
            if random.random() < self._threshold:
                tensor = tensor.flip(ax)
        return tensor

    def __repr__(self):
        return f"{self.__class__.__name__}" + (f" with axis={self._axis}." if self._axis else "")
