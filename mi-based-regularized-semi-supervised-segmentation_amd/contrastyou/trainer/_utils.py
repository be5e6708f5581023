"""Cluster heads on the MI355X kernels (ref: contrastyou/trainer/_utils.py:96-168).

``ClusterHead`` (global: avg-pool -> Linear -> softmax/T) and ``LocalClusterHead`` (per pixel:
1x1 conv -> channel softmax/T), each with ``num_subheads`` independent sub-heads, same constructor
arguments and the reference's state_dict keys (``_headers.<s>.<idx>.{weight,bias}``).  All sub-heads
of a head run in ONE fused launch.  ``forward(features)`` keeps the reference signature and returns a
list of per-sub-head probabilities; ``forward_gathered(features, src, flips)`` additionally fuses the
epocher's sample gather / flip replay / cat (semi_seg/epocher.py:258-273) into the same kernel.
``head_type='linear'`` with ``normalize=False`` (the shipped config, config/semi.yaml:45-55) runs on the tuned
kernels of csrc/heads.hip / mi_global.hip; ``head_type='mlp'`` and ``normalize=True`` (ref :106-126, :146-161) run on the
generic fused kernels of csrc/heads_var.hip (same gather / flip fusion, forward recomputed in the backward).
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


class Normalize(nn.Module):
    """Parameter-free place holder (ref _utils.py:26-33) so the Sequential indices -- hence the state_dict keys -- match."""

    def forward(self, input):
        return torch.nn.functional.normalize(input, p=2, dim=1)


_GLOBAL_HIDDEN = 128       # ref _utils.py:120: the pooled mlp head's hidden width is fixed


def _stack(layers, attr):
    return stacked_param([getattr(layer, attr) for layer in layers])


class _SubHeadParams:
    """The S sub-heads' parameters as stacked [S, ...] tensors (views of the flat buffers when the optimiser laid them out back
    to back): first layer, and second layer for head_type='mlp'."""

    def _register(self, first, second):
        self._first, self._second = first, second
        for layers in ((first,) if second is None else (first, second)):
            register_adjacent([layer.weight for layer in layers])
            register_adjacent([layer.bias for layer in layers])

    def _stacked(self):
        def flat2(w):       # conv weights [S, R, C, 1, 1] -> [S, R, C], keeping the flat-slot bookkeeping of the stack
            v = w.view(w.shape[0], w.shape[1], -1)
            if hasattr(w, "_miseg_stack_params"):
                v._miseg_stack_params = w._miseg_stack_params
            return v
        w1, b1 = flat2(_stack(self._first, "weight")), _stack(self._first, "bias")
        if self._second is None:
            return w1, b1, None, None
        return w1, b1, flat2(_stack(self._second, "weight")), _stack(self._second, "bias")


class ClusterHead(nn.Module, _SubHeadParams):
    def __init__(self, input_dim, num_clusters=5, num_subheads=10, head_type="linear", T=1, normalize=False) -> None:
        super().__init__()
        assert head_type in ("linear", "mlp"), head_type
        self._input_dim, self._num_clusters, self._num_subheads, self._T, self._normalize = \
            input_dim, num_clusters, num_subheads, T, normalize
        self._head_type = head_type
        tail = lambda: [Normalize() if normalize else Identical(), SoftmaxWithT(1, T=T)]  # noqa: E731
        if head_type == "linear":
            build = lambda: nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), Flatten(), nn.Linear(input_dim, num_clusters), *tail())  # noqa: E731
        else:
            build = lambda: nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), Flatten(), nn.Linear(input_dim, _GLOBAL_HIDDEN),  # noqa: E731
                                          nn.LeakyReLU(0.01, inplace=True), nn.Linear(_GLOBAL_HIDDEN, num_clusters), *tail())
        self._headers = nn.ModuleList([build() for _ in range(num_subheads)])
        self._register([h[2] for h in self._headers], [h[4] for h in self._headers] if head_type == "mlp" else None)

    def forward_gathered(self, features: Tensor, src: Tensor) -> Tensor:
        w1, b1, w2, b2 = self._stacked()
        if self._head_type == "linear" and not self._normalize:
            return ops.global_head(features, w1, b1, src, self._T)          # [S, M, K] -- the shipped, tuned kernels
        return ops.global_head_var(features, w1, b1, w2, b2, src, self._T, self._normalize)

    def forward(self, features: Tensor) -> List[Tensor]:
        src = torch.arange(features.shape[0], dtype=torch.int32, device=features.device)
        return list(self.forward_gathered(features, src))


class LocalClusterHead(nn.Module, _SubHeadParams):
    def __init__(self, input_dim, head_type="linear", num_clusters=10, num_subheads=10, T=1, interm_dim=64,
                 normalize=False) -> None:
        super().__init__()
        assert head_type in ("linear", "mlp"), head_type
        self._T, self._normalize, self._head_type = T, normalize, head_type
        tail = lambda: [Normalize() if normalize else Identical(), SoftmaxWithT(1, T=T)]  # noqa: E731
        if head_type == "linear":
            build = lambda: nn.Sequential(nn.Conv2d(input_dim, num_clusters, 1, 1, 0), *tail())  # noqa: E731
        else:
            build = lambda: nn.Sequential(nn.Conv2d(input_dim, interm_dim, 1, 1, 0), nn.LeakyReLU(0.01, inplace=True),  # noqa: E731
                                          nn.Conv2d(interm_dim, num_clusters, 1, 1, 0), *tail())
        self._headers = nn.ModuleList([build() for _ in range(num_subheads)])
        self._register([h[0] for h in self._headers], [h[2] for h in self._headers] if head_type == "mlp" else None)

    def forward_gathered(self, features: Tensor, src: Tensor, flips: Optional[Tensor]) -> Tensor:
        w1, b1, w2, b2 = self._stacked()
        if self._head_type == "linear" and not self._normalize:
            return ops.local_head(features, w1, b1, src, flips, self._T)      # [S, M, K, H, W] -- the shipped, tuned kernels
        return ops.local_head_var(features, w1, b1, w2, b2, src, flips, self._T, self._normalize)

    def forward(self, features: Tensor) -> List[Tensor]:
        src = torch.arange(features.shape[0], dtype=torch.int32, device=features.device)
        return list(self.forward_gathered(features, src, None))
