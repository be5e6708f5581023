"""KL_div / Entropy (ref whl:deepclustering2/loss/kl_losses.py:20-49, 76-141).

``KL_div()(prob, target)`` keeps the reference call form.  On the train-step hot path the epocher calls
``KL_div.from_logits(logits, labels)`` instead, which is the fused HIP kernel (softmax + one-hot + KL +
mean, and its backward) -- numerically the same expression evaluated in one pass.
"""
from __future__ import annotations

from typing import List, Optional, Union

import torch
from torch import Tensor, nn

from deepclustering2.utils import simplex


class KL_div(nn.Module):
    def __init__(self, reduction="mean", eps=1e-16, weight: Union[List[float], Tensor] = None, verbose=True):
        super().__init__()
        assert reduction in ("mean", "sum", "none"), reduction
        self._eps, self._reduction = eps, reduction
        self._weight: Optional[Tensor] = None
        if weight is not None:
            w = torch.as_tensor(weight).float()
            self._weight = w / w.sum() * len(w)

    def forward(self, prob: Tensor, target: Tensor, **kwargs) -> Tensor:
        if not kwargs.get("disable_assert"):
            assert prob.shape == target.shape
            assert simplex(prob), prob
            assert simplex(target), target
            assert not target.requires_grad
            assert prob.requires_grad
        kl = -target * torch.log((prob + self._eps) / (target + self._eps))
        if self._weight is not None:
            shape = [1, -1] + [1] * (kl.dim() - 2)
            kl = kl * self._weight.to(kl.device).view(*shape)
        kl = kl.sum(1)
        return kl.mean() if self._reduction == "mean" else kl.sum() if self._reduction == "sum" else kl

    def supports_fused(self) -> bool:
        return self._weight is None and self._reduction == "mean" and self._eps == 1e-16

    def from_logits(self, logits: Tensor, labels: Tensor) -> Tensor:
        """== self(softmax(logits, 1), class2one_hot(labels)) in one fused HIP kernel (+ fused backward)."""
        from miseg_amd import ops
        assert self.supports_fused()
        return ops.softmax_kl(logits, labels)

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        sd["weight"], sd["reduction"] = self._weight, self._reduction
        return sd

    def load_state_dict(self, state_dict, *args, **kwargs):
        self._reduction, self._weight = state_dict["reduction"], state_dict["weight"]


class Entropy(nn.Module):
    def __init__(self, reduction="mean", eps=1e-16):
        super().__init__()
        self._eps, self._reduction = eps, reduction

    def forward(self, input: Tensor) -> Tensor:
        assert simplex(input)
        e = -(input * (input + self._eps).log()).sum(1)
        return e.mean() if self._reduction == "mean" else e.sum() if self._reduction == "sum" else e
