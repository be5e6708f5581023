"""Oracle: global and local IIC mutual-information losses.

Restates ``contrastyou/losses/iic_loss.py`` (IIDLoss :31-71, compute_joint :74-94,
IIDSegmentationLoss :97-149, patch_generator :152-160,
IIDSegmentationSmallPathLoss :164-189) with explicit shifts / einsums instead of
the reference's conv2d-as-correlation trick, plus the closed-form gradient the
HIP backward kernels implement (checked against autograd in the tests).

Test infrastructure only (see oracle/__init__.py).
"""
from __future__ import annotations

import numpy as np
import torch
from torch import Tensor

GLOBAL_EPS = 1e-10  # literal inside every log of IIDLoss (iic_loss.py:65-69)
LOCAL_EPS = 1e-16  # literal of IIDSegmentationLoss (iic_loss.py:124,141-143)


def simplex(t: Tensor, axis: int = 1) -> bool:
    """whl:deepclustering2/utils/general.py:176-185 -- channel sums within 1e-4 of 1."""
    s = t.sum(axis).type(torch.float32)
    return bool(torch.allclose(s, torch.ones_like(s), rtol=1e-4, atol=1e-4))


# --------------------------------------------------------------------------- global MI
def global_joint(x: Tensor, y: Tensor, symmetric: bool = True) -> Tensor:
    """compute_joint (iic_loss.py:74-94): P = sum_n x_n (outer) y_n, symmetrised, normalised."""
    assert x.dim() == 2 and x.shape == y.shape
    p = torch.einsum("ni,nj->ij", x, y)
    if symmetric:
        p = (p + p.t()) / 2.0
    return p / p.sum()


def iid_loss(x: Tensor, y: Tensor, lamb: float = 1.0):
    """IIDLoss.forward (iic_loss.py:43-71) -> (loss, loss_no_lamb, P[K,K]).

    Marginals are taken from the symmetrised joint: p_i = row sums, p_j = column
    sums (iic_loss.py:56-59).
    """
    assert simplex(x) and simplex(y)
    p = global_joint(x, y)
    k = p.shape[0]
    p_i = p.sum(1).view(k, 1)
    p_j = p.sum(0).view(1, k)
    log_p = torch.log(p + GLOBAL_EPS)
    log_i = torch.log(p_i + GLOBAL_EPS)
    log_j = torch.log(p_j + GLOBAL_EPS)
    loss = -(p * (log_p - lamb * log_j - lamb * log_i)).sum()
    loss_no_lamb = -(p * (log_p - log_j - log_i)).sum()
    return loss, loss_no_lamb, p


# --------------------------------------------------------------------------- local MI
def local_joint_raw(x: Tensor, y: Tensor, padding: int) -> Tensor:
    """Raw displacement joint R[dy, dx, i, j] (T x T x K x K, T = 2*padding+1).

    Same numbers as ``F.conv2d(x.permute(1,0,2,3), weight=y.permute(1,0,2,3),
    padding=p)`` at iic_loss.py:120-123, re-indexed T,T,K,K:
    ``R[a,b,i,j] = sum_{n,h,w} Xpad[n,i,h+a,w+b] * Y[n,j,h,w]`` with zero padding
    of X by ``padding`` on every side (a,b = 0..2p; displacement = a-p, b-p).
    """
    n, k, h, w = x.shape
    p = int(padding)
    t = 2 * p + 1
    xpad = torch.nn.functional.pad(x, (p, p, p, p))
    out = x.new_zeros(t, t, k, k)
    for a in range(t):
        for b in range(t):
            win = xpad[:, :, a:a + h, b:b + w]
            out[a, b] = torch.einsum("nihw,njhw->ij", win, y)
    return out


def local_mi_from_raw(raw: Tensor, lamda: float = 1.0) -> Tensor:
    """Epilogue of IIDSegmentationLoss (iic_loss.py:124-146) on R[T,T,K,K].

    global-min shift (detached) + 1e-16; per-displacement normalise; symmetrise
    the KxK part; marginals; ``-sum P (log P - lam log Pi - lam log Pj) / T^2``.
    """
    t = raw.shape[0]
    s = raw - raw.min().detach() + LOCAL_EPS
    q = s / s.sum(dim=3, keepdim=True).sum(dim=2, keepdim=True)
    ps = (q + q.transpose(2, 3)) / 2.0
    col = ps.sum(dim=2, keepdim=True)  # p_i_mat: function of j (iic_loss.py:135)
    row = ps.sum(dim=3, keepdim=True)  # p_j_mat: function of i (iic_loss.py:136)
    terms = ps * (torch.log(ps + LOCAL_EPS) - lamda * torch.log(col + LOCAL_EPS)
                  - lamda * torch.log(row + LOCAL_EPS))
    return -terms.sum() / float(t * t)


def local_mi_grad_wrt_raw(raw: Tensor, lamda: float = 1.0) -> Tensor:
    """Closed-form dLoss/dR for :func:`local_mi_from_raw` (min shift detached).

    This is the formula the HIP epilogue kernel emits; tests check it against
    autograd through local_mi_from_raw in fp64.
    """
    t = raw.shape[0]
    eps = LOCAL_EPS
    s = raw - raw.min() + eps
    z = s.sum(dim=(2, 3), keepdim=True)
    q = s / z
    ps = (q + q.transpose(2, 3)) / 2.0
    col = ps.sum(dim=2, keepdim=True)
    row = ps.sum(dim=3, keepdim=True)
    gs = -(torch.log(ps + eps) + ps / (ps + eps)
           - lamda * (torch.log(col + eps) + col / (col + eps))
           - lamda * (torch.log(row + eps) + row / (row + eps))) / float(t * t)
    gq = (gs + gs.transpose(2, 3)) / 2.0
    return (gq - (gq * q).sum(dim=(2, 3), keepdim=True)) / z


def iid_seg_loss(x: Tensor, y: Tensor, padding: int, lamda: float = 1.0, mask: Tensor | None = None) -> Tensor:
    """IIDSegmentationLoss.__call__ (iic_loss.py:107-149)."""
    assert x.shape == y.shape
    assert simplex(x)
    if mask is not None:
        x = x * mask
        y = y * mask
    loss = local_mi_from_raw(local_joint_raw(x, y, padding), lamda)
    if torch.isnan(loss):
        raise RuntimeError(loss)
    return loss


def patch_origins(extent: int, patch: int, step: int) -> list[int]:
    """Window origins of patch_generator (iic_loss.py:152-160) along one axis.

    ``arange(0, extent - patch, step)`` then append ``max(extent - patch, 0)``
    (the clamped last window; it duplicates nothing only when not on the grid).
    """
    base = list(np.arange(0, extent - patch, step))
    base.append(max(extent - patch, 0))
    return [int(v) for v in base]


def patch_windows(h: int, w: int, patch: tuple[int, int], step: tuple[int, int]):
    """All (h0, h1, w0, w1) windows in the reference's iteration order."""
    out = []
    for h0 in patch_origins(h, patch[0], step[0]):
        for w0 in patch_origins(w, patch[1], step[1]):
            out.append((h0, min(h0 + patch[0], h), w0, min(w0 + patch[1], w)))
    return out


def iid_seg_small_patch_loss(x: Tensor, y: Tensor, padding: int, patch_size: int, lamda: float = 1.0,
                             mask: Tensor | None = None) -> Tensor:
    """IIDSegmentationSmallPathLoss.__call__ (iic_loss.py:173-186): mean over patches,
    step = patch_size // 2 (iic_loss.py:169-171)."""
    assert x.shape == y.shape
    ps = (patch_size, patch_size)
    st = (patch_size // 2, patch_size // 2)
    losses = []
    for (h0, h1, w0, w1) in patch_windows(x.shape[2], x.shape[3], ps, st):
        m = None if mask is None else mask[:, :, h0:h1, w0:w1]
        losses.append(iid_seg_loss(x[:, :, h0:h1, w0:w1], y[:, :, h0:h1, w0:w1], padding, lamda, m))
    return sum(losses) / float(len(losses))  # average_iter, contrastyou/helper/utils.py:46-47
