"""Global and local IIC mutual-information losses on the MI355X kernels.

Drop-in for ref ``contrastyou/losses/iic_loss.py``: same class names, constructor arguments, return
values, assertions (``AssertionError`` on non-simplex / shape / requires_grad) and the ``RuntimeError``
on a NaN loss.  The arithmetic runs in ``miseg_amd`` (csrc/mi_local.hip, csrc/mi_global.hip):
the K x K x T x T displacement joint is one MFMA contraction with both displacement axes stacked into
the tile dimensions, the min-shift / normalise / symmetrise / MI epilogue is one small kernel, and all
patches of ``IIDSegmentationSmallPathLoss`` go through a single batched launch.
"""
from __future__ import annotations

import sys
from itertools import repeat
from typing import Iterable

import numpy as np
import torch
from torch import Tensor, nn

from contrastyou.helper import average_iter
from miseg_amd import checks, ops

__all__ = ["IIDLoss", "compute_joint", "IIDSegmentationLoss", "IIDSegmentationSmallPathLoss", "patch_generator"]


def simplex(t: Tensor, axis: int = 1) -> bool:
    """Channel sums within 1e-4 (rtol) + 1e-4 (atol) of one (whl:deepclustering2/utils/general.py:176-185).  Returns a
    host bool, i.e. one device sync; the losses below use the non-blocking ``checks.assert_simplex`` instead."""
    return int(checks.simplex_violations(t, axis)) == 0


def _pair(x):
    return tuple(x) if isinstance(x, Iterable) else tuple(repeat(x, 2))


class IIDLoss(nn.Module):
    """ref iic_loss.py:31-71.  ``forward(x[N,K], y[N,K]) -> (loss, loss_no_lamb, p_i_j[K,K])``."""

    def __init__(self, lamb: float = 1.0, eps: float = sys.float_info.epsilon):
        super().__init__()
        self.lamb = float(lamb)
        self.eps = float(eps)  # unused by the reference too: the literal 1e-10 sits inside the logs

    def forward(self, x_out: Tensor, x_tf_out: Tensor):
        checks.assert_simplex(x_out, 1, "x_out not normalized.")
        checks.assert_simplex(x_tf_out, 1, "x_tf_out not normalized.")
        loss, loss_no_lamb, joint = ops.global_mi(x_out.unsqueeze(0), x_tf_out.unsqueeze(0), self.lamb)
        return loss[0], loss_no_lamb[0], joint[0]


def compute_joint(x_out: Tensor, x_tf_out: Tensor, symmetric: bool = True) -> Tensor:
    """ref iic_loss.py:74-94: normalised sum_n x_n (outer) y_n, symmetrised unless ``symmetric=False``; differentiable."""
    checks.assert_simplex(x_out, 1, "x_out not normalized.")
    checks.assert_simplex(x_tf_out, 1, "x_tf_out not normalized.")
    bn, k = x_out.shape
    assert x_tf_out.size(0) == bn and x_tf_out.size(1) == k
    return ops.global_joint(x_out.unsqueeze(0), x_tf_out.unsqueeze(0), symmetric)[0]


def _windows(h: int, w: int, patch, step):
    """Window list of patch_generator (ref iic_loss.py:152-160), same iteration order."""
    def origins(extent, p, s):
        o = list(np.arange(0, extent - p, s))
        o.append(max(extent - p, 0))
        return [int(v) for v in o]
    return [(h0, min(h0 + patch[0], h), w0, min(w0 + patch[1], w)) for h0 in origins(h, patch[0], step[0])
            for w0 in origins(w, patch[1], step[1])]


def patch_generator(feature_map: Tensor, patch_size=(32, 32), step_size=(16, 16)):
    """Yields the overlapping crops the reference yields (views, no copies)."""
    _, _, h, w = feature_map.shape
    for h0, h1, w0, w1 in _windows(h, w, patch_size, step_size):
        yield feature_map[:, :, h0:h1, w0:w1]


class IIDSegmentationLoss(nn.Module):
    """ref iic_loss.py:97-149.  ``__call__(x[N,K,H,W], y, mask=None) -> scalar``."""

    def __init__(self, lamda=1.0, padding=7, eps: float = sys.float_info.epsilon) -> None:
        super().__init__()
        self.lamda = lamda
        self.padding = padding
        self.eps = eps

    def _check(self, x_out: Tensor, x_tf_out: Tensor, mask):
        assert x_out.requires_grad and x_tf_out.requires_grad
        if mask is not None:
            assert not mask.requires_grad
        checks.assert_simplex(x_out, 1, "x_out not normalized.")
        assert x_out.shape == x_tf_out.shape

    def forward(self, x_out: Tensor, x_tf_out: Tensor, mask: Tensor = None) -> Tensor:
        self._check(x_out, x_tf_out, mask)
        h, w = x_out.shape[2:]
        loss = ops.local_mi_losses(x_out, x_tf_out, self.padding, [(0, h, 0, w)], self.lamda, mask)[0]
        checks.raise_if_nan(loss, "IIDSegmentationLoss is nan")
        return loss


class IIDSegmentationSmallPathLoss(IIDSegmentationLoss):
    """ref iic_loss.py:164-189: mean of IIDSegmentationLoss over overlapping patches (stride = patch/2)."""

    def __init__(self, lamda=1.0, padding=7, eps: float = sys.float_info.epsilon, patch_size=32) -> None:
        super().__init__(lamda, padding, eps)
        self._patch_size = _pair(patch_size)
        self._step_size = _pair(patch_size // 2) if not isinstance(patch_size, Iterable) else tuple(p // 2 for p in patch_size)

    def forward(self, x_out: Tensor, x_tf_out: Tensor, mask: Tensor = None):
        assert x_out.shape == x_tf_out.shape, (x_out.shape, x_tf_out.shape)
        self._check(x_out, x_tf_out, mask)
        h, w = x_out.shape[2:]
        wins = _windows(h, w, self._patch_size, self._step_size)
        losses = ops.local_mi_losses(x_out, x_tf_out, self.padding, wins, self.lamda, mask)
        checks.raise_if_nan(losses, "IIDSegmentationSmallPathLoss: a patch loss is nan")
        return average_iter(list(losses)) if len(wins) <= 4 else losses.sum() / float(len(wins))

    def forward_heads(self, probs: Tensor, ub: int, mask: Tensor = None, lazy: bool = False):
        """``[self(p[:ub], p[ub:]) for p in probs]`` as one autograd node: probs[S, 2*UB, K, H, W] -> loss[S].
        Same checks and values as S separate calls (ref semi_seg/epocher.py:264-272 loops the sub-heads)."""
        assert probs.requires_grad and probs.dim() == 5 and probs.shape[1] == 2 * ub, probs.shape
        if mask is not None:
            assert not mask.requires_grad
        checks.assert_simplex(probs, 2, "probs not normalized.")
        h, w = probs.shape[3:]
        wins = _windows(h, w, self._patch_size, self._step_size)
        losses = ops.local_mi_heads(probs, ub, self.padding, wins, self.lamda, mask)   # [S, P]
        checks.raise_if_nan(losses, "IIDSegmentationSmallPathLoss: a patch loss is nan")
        if lazy:   # mean over sub-heads of the mean over patches, kept symbolic (miseg_amd.lazy): no reduction kernels
            from miseg_amd.lazy import LinearLoss
            return LinearLoss.mean(losses)
        return losses.sum(1) / float(len(wins))

    def __repr__(self):
        return f"{self.__class__.__name__} with patch_size={self._patch_size} and padding={self.padding}."
