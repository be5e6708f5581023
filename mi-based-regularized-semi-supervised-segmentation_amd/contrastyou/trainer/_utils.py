"""Cluster heads on the MI355X kernels (ref: contrastyou/trainer/_utils.py:96-168).

``ClusterHead`` (global: avg-pool -> Linear -> softmax/T) and ``LocalClusterHead`` (per pixel:
1x1 conv -> channel softmax/T), each with ``num_subheads`` independent sub-heads, same constructor
arguments and the reference's state_dict keys (``_headers.<s>.<idx>.{weight,bias}``).  All sub-heads
of a head run in ONE fused launch.  ``forward(features)`` keeps the reference signature and returns a
list of per-sub-head probabilities; ``forward_gathered(features, src, flips)`` additionally fuses the
epocher's sample gather / flip replay / cat (semi_seg/epocher.py:258-273) into the same kernel.
Only ``head_type='linear'`` with ``normalize=False`` (the shipped config, config/semi.yaml:45-55) is on
the hot path; other variants raise NotImplementedError.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import Tensor, nn

from miseg_amd import ops
from miseg_amd.gradslot import register_adjacent, stacked_param


class Flatten(nn.Module):
    def forward(self, features):
        return features.view(features.shape[0], -1)


class Identical(nn.Module):
    def forward(self, input):
        return input


class SoftmaxWithT(nn.Softmax):
    def __init__(self, dim, T: float = 0.1) -> None:
        super().__init__(dim)
        self._T = T


def _check_variant(head_type, normalize):
    assert head_type in ("linear", "mlp"), head_type
    if head_type != "linear" or normalize:
        raise NotImplementedError("only head_type='linear', normalize=False is implemented on the MI355X hot path")


class ClusterHead(nn.Module):
    def __init__(self, input_dim, num_clusters=5, num_subheads=10, head_type="linear", T=1, normalize=False) -> None:
        super().__init__()
        _check_variant(head_type, normalize)
        self._input_dim, self._num_clusters, self._num_subheads, self._T, self._normalize = \
            input_dim, num_clusters, num_subheads, T, normalize
        self._headers = nn.ModuleList([
            nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), Flatten(), nn.Linear(input_dim, num_clusters), Identical(),
                          SoftmaxWithT(1, T=T)) for _ in range(num_subheads)])
        register_adjacent([h[2].weight for h in self._headers])   # flat buffers keep the S sub-heads back to back
        register_adjacent([h[2].bias for h in self._headers])

    def _wb(self):
        return stacked_param([h[2].weight for h in self._headers]), stacked_param([h[2].bias for h in self._headers])

    def forward_gathered(self, features: Tensor, src: Tensor) -> Tensor:
        w, b = self._wb()
        return ops.global_head(features, w, b, src, self._T)            # [S, M, K]

    def forward(self, features: Tensor) -> List[Tensor]:
        src = torch.arange(features.shape[0], dtype=torch.int32, device=features.device)
        return list(self.forward_gathered(features, src))


class LocalClusterHead(nn.Module):
    def __init__(self, input_dim, head_type="linear", num_clusters=10, num_subheads=10, T=1, interm_dim=64,
                 normalize=False) -> None:
        super().__init__()
        _check_variant(head_type, normalize)
        self._T, self._normalize = T, normalize
        self._headers = nn.ModuleList([
            nn.Sequential(nn.Conv2d(input_dim, num_clusters, 1, 1, 0), Identical(), SoftmaxWithT(1, T=T))
            for _ in range(num_subheads)])
        register_adjacent([h[0].weight for h in self._headers])
        register_adjacent([h[0].bias for h in self._headers])

    def _wb(self):
        w = stacked_param([h[0].weight for h in self._headers])          # [S, K, C, 1, 1]
        wv = w.view(w.shape[0], w.shape[1], -1)
        if hasattr(w, "_miseg_stack_params"):
            wv._miseg_stack_params = w._miseg_stack_params
        return wv, stacked_param([h[0].bias for h in self._headers])

    def forward_gathered(self, features: Tensor, src: Tensor, flips: Optional[Tensor]) -> Tensor:
        w, b = self._wb()
        return ops.local_head(features, w, b, src, flips, self._T)      # [S, M, K, H, W]

    def forward(self, features: Tensor) -> List[Tensor]:
        src = torch.arange(features.shape[0], dtype=torch.int32, device=features.device)
        return list(self.forward_gathered(features, src, None))
