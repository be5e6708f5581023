"""Oracle: supervised KL, UDA MSE, one-hot helpers, flip replay, Dice, Adam, lr schedule.

Restates (whl = the reference's vendored deepclustering2 wheel):
* KL_div.forward                whl:deepclustering2/loss/kl_losses.py:107-126
* class2one_hot / one_hot / sset whl:deepclustering2/utils/general.py:145-163,188-196,221-235
* nn.MSELoss on two softmaxes    semi_seg/epocher.py:221-224, semi_seg/trainer.py:194
* TensorRandomFlip + FixRandomSeed whl:deepclustering2/augment/tensor_augment.py:31-39,
                                 whl:deepclustering2/decorator/decorator.py:196-212
* UniversalDice                  whl:deepclustering2/meters2/individual_meters/general_dice_meter.py:41-175
* Adam (L2 weight decay)         torch.optim.Adam as configured at semi_seg/trainer.py:67-72,179-184
* warm-up + cosine schedule      whl:deepclustering2/schedulers/warmup_scheduler.py:8-75 around
                                 CosineAnnealingLR(T_max=max_epoch-warmup, eta_min=1e-7), semi_seg/trainer.py:52-65
Test infrastructure only.
"""
from __future__ import annotations

import math
import random

import numpy as np
import torch
from torch import Tensor

KL_EPS = 1e-16


def class2one_hot(seg: Tensor, num_classes: int) -> Tensor:
    """int64 one-hot along dim 1 (general.py:221-235); labels must lie in [0, C)."""
    if seg.dim() == 2:
        seg = seg.unsqueeze(0)
    vals = set(int(v) for v in seg.unique())
    assert vals.issubset(set(range(num_classes))), vals
    return torch.stack([seg == c for c in range(num_classes)], dim=1).type(torch.long)


def is_one_hot(t: Tensor, axis: int = 1) -> bool:
    s = t.sum(axis).type(torch.float32)
    ok = bool(torch.allclose(s, torch.ones_like(s), rtol=1e-4, atol=1e-4))
    return ok and set(float(v) for v in t.unique()).issubset({0.0, 1.0})


def kl_div(prob: Tensor, target: Tensor, reduction: str = "mean") -> Tensor:
    """``sum_c -t * log((p+eps)/(t+eps))`` then mean over (b,h,w) (kl_losses.py:113-126)."""
    assert prob.shape == target.shape
    kl = (-target * torch.log((prob + KL_EPS) / (target + KL_EPS))).sum(1)
    return kl.mean() if reduction == "mean" else kl.sum() if reduction == "sum" else kl


def softmax_mse(logits_a: Tensor, logits_b: Tensor) -> Tensor:
    """MSELoss(softmax(a), softmax(b).detach()) -- mean over ALL elements (epocher.py:221-224)."""
    return ((logits_a.softmax(1) - logits_b.softmax(1).detach()) ** 2).mean()


def flip_decisions(seed: int, batch: int, threshold: float = 0.8, num_axes: int = 2) -> list[list[bool]]:
    """Decisions TensorRandomFlip(axis=[1,2], threshold) makes for a batch under FixRandomSeed(seed):
    ``random.seed(seed)`` then, per sample in order, one ``random.random() < threshold`` draw per
    axis (H first, then W).  The caller's RNG state is restored, as the context manager does."""
    state, npstate = random.getstate(), np.random.get_state()
    try:
        np.random.seed(seed)
        random.seed(seed)
        return [[random.random() < threshold for _ in range(num_axes)] for _ in range(batch)]
    finally:
        np.random.set_state(npstate)
        random.setstate(state)


def apply_flips(x: Tensor, decisions) -> Tensor:
    """Per-sample clone + flips; axis 1 of a [C,H,W] sample is H, axis 2 is W."""
    out = []
    for sample, (fh, fw) in zip(x, decisions):
        s = sample.clone()
        if fh:
            s = s.flip(1)
        if fw:
            s = s.flip(2)
        out.append(s)
    return torch.stack(out, dim=0)


# ----------------------------------------------------------------------------- dice
def dice_counts(pred: Tensor, target: Tensor, num_classes: int):
    """Per-sample integer intersections / unions [B,C] (general_dice_meter.py:141-172)."""
    p = class2one_hot(pred, num_classes)
    t = class2one_hot(target, num_classes)
    dims = list(range(2, p.dim()))
    return (p * t).sum(dims), (p + t).sum(dims)


class DiceMeter:
    """UniversalDice: per-group (patient) 3D dice ``(2*I+1e-6)/(U+1e-6)``, mean over groups,
    DSC_mean = mean over the reported classes 1..C-1 (general_dice_meter.py:95-122)."""

    def __init__(self, num_classes: int, report_axis=None):
        self.c = num_classes
        self.report = list(range(num_classes)) if report_axis is None else list(report_axis)
        self.inter, self.union, self.names, self.n = [], [], [], 0

    def add(self, pred: Tensor, target: Tensor, group_name=None):
        i, u = dice_counts(pred, target, self.c)
        b = pred.shape[0]
        names = [f"{self.n}_{k:03d}" for k in range(b)]
        if group_name is not None:
            names = [group_name] * b if isinstance(group_name, str) else list(group_name)
        self.inter.append(i)
        self.union.append(u)
        self.names.extend(names)
        self.n += 1

    def summary(self) -> dict:
        if self.n == 0:
            means = [float("nan")] * self.c
        else:
            inter, union = torch.cat(self.inter, 0), torch.cat(self.union, 0)
            names = np.asarray(self.names)
            rows = []
            for g in sorted(set(self.names)):
                idx = torch.from_numpy(names == g)
                rows.append((2 * inter[idx].sum(0) + 1e-6) / (union[idx].sum(0) + 1e-6))
            means = torch.stack(rows, 0).mean(0)
        rep = {f"DSC{i}": float(means[i]) for i in self.report}
        rep["DSC_mean"] = sum(rep.values()) / len(rep)
        return rep


# ----------------------------------------------------------------------------- optimiser
def adam_step(params, grads, exp_avg, exp_avg_sq, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8,
              weight_decay=0.0):
    """One torch.optim.Adam update, in place (``step`` is the 1-based count after this update)."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        if weight_decay != 0.0:
            g = g + weight_decay * p
        m.mul_(beta1).add_(g, alpha=1.0 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


def warmup_cosine_lrs(base_lr: float, multiplier: float, warmup_max: int, max_epoch: int, eta_min: float = 1e-7):
    """lr used during epoch e = 0..max_epoch-1 when ``scheduler.step()`` is called once per epoch
    (semi_seg/trainer.py:52-65,98-99).

    Warm-up: ``base*((mult-1)*e/warm+1)`` for e <= warm (warmup_scheduler.py:37-41).  On the first
    step past warm-up the wrapper rescales the cosine scheduler's base_lrs and returns its
    ``get_lr()`` while that scheduler still sits at last_epoch=0 (warmup_scheduler.py:27-35); with
    the chainable CosineAnnealingLR of the torch build the goldens were made with (2.10) the
    recursive form is then evaluated from ``lr=top`` at t=0, so the whole cosine leg is the closed
    form scaled by ``2/(1+cos(pi/T))`` and indexed t = e - warm - 1.  (Pinned by the golden vector;
    the reference pins nothing here -- SURVEY.md 8(c).)"""
    lrs = []
    t_max = max_epoch - warmup_max
    top = base_lr * multiplier
    for e in range(max_epoch):
        if e <= warmup_max:
            lrs.append(base_lr * ((multiplier - 1.0) * e / warmup_max + 1.0))
        else:
            t = e - warmup_max - 1
            lrs.append(eta_min + (top - eta_min) * (1 + math.cos(math.pi * t / t_max)) / (1 + math.cos(math.pi / t_max)))
    return lrs
