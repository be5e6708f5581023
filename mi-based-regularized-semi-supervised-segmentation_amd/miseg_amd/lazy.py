"""Scalar loss arithmetic without kernels.

The reference combines its loss terms with Python scalar arithmetic on 0-d tensors (means over sub-heads and patches,
importance-weighted averages, ``cons_weight * uda + iic_weight * iic``, ``sup + reg_weight * reg``; ref
semi_seg/epocher.py:165-176, 258-284, 318-323): ~60 one-element kernels per iteration forward and backward.  Every one
of those expressions is LINEAR in a handful of kernel outputs with coefficients known on the host, so here they are kept
symbolic: a ``LinearLoss`` is ``sum_i coeff_i * sum(term_i)``.  ``backward()`` seeds autograd directly at the kernel
outputs with constant gradient tensors (cached per shape/coefficient), and all reported scalars of an iteration are
produced by ONE concatenation and ONE matrix-vector product against a cached coefficient matrix (``evaluate``).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple, Union

import torch
from torch import Tensor

Number = Union[int, float]
_CONST: Dict[tuple, Tensor] = {}


def _const(key, build) -> Tensor:
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = build()
    return t


class LinearLoss:
    __slots__ = ("terms",)

    def __init__(self, terms: List[Tuple[Tensor, float]]):
        self.terms = terms            # [(tensor, coefficient of sum(tensor))]

    # ---- construction
    @staticmethod
    def of(x) -> "LinearLoss":
        if isinstance(x, LinearLoss):
            return x
        if isinstance(x, Tensor):
            return LinearLoss([(x, 1.0)])
        raise TypeError(f"cannot build a LinearLoss from {type(x).__name__}")

    @staticmethod
    def mean(t: Tensor) -> "LinearLoss":
        return LinearLoss([(t, 1.0 / float(t.numel()))])

    # ---- arithmetic with numbers, tensors and other LinearLoss objects
    def __add__(self, other):
        if isinstance(other, (int, float)):
            if other == 0:
                return self
            raise TypeError("LinearLoss + non-zero constant is not supported")
        return LinearLoss(self.terms + LinearLoss.of(other).terms)

    __radd__ = __add__

    def __sub__(self, other):
        return self + (-LinearLoss.of(other))

    def __neg__(self):
        return LinearLoss([(t, -c) for t, c in self.terms])

    def __mul__(self, k: Number):
        if not isinstance(k, (int, float)):
            return NotImplemented
        return LinearLoss([(t, c * float(k)) for t, c in self.terms])

    __rmul__ = __mul__

    def __truediv__(self, k: Number):
        if not isinstance(k, (int, float)):
            return NotImplemented
        return LinearLoss([(t, c / float(k)) for t, c in self.terms])

    # ---- use
    def merged(self) -> List[Tuple[Tensor, float]]:
        acc: Dict[int, List] = {}
        for t, c in self.terms:
            e = acc.get(id(t))
            if e is None:
                acc[id(t)] = [t, c]
            else:
                e[1] += c
        return [(t, c) for t, c in acc.values()]

    def backward(self) -> None:
        roots, grads = [], []
        for t, c in self.merged():
            if t.requires_grad:
                roots.append(t)
                grads.append(_const(("g", tuple(t.shape), t.dtype, str(t.device), c),
                                    lambda t=t, c=c: torch.full(t.shape, c, dtype=t.dtype, device=t.device)))
        torch.autograd.backward(roots, grads)

    def detach(self) -> "LinearLoss":
        return LinearLoss([(t.detach(), c) for t, c in self.terms])

    def value(self) -> Tensor:
        return evaluate([self])[0]


def evaluate(items: Sequence[Union[LinearLoss, Tensor]], passthrough: Sequence[Tensor] = ()) -> Tensor:
    """float32 device vector with the value of every item (one cat of the distinct base tensors, one mat-vec), followed by the
    0-d ``passthrough`` tensors verbatim.  The mat-vec multiplies EVERY base value into EVERY row (with coefficient 0 where a
    row does not use it), and 0 * NaN = 0 * inf = NaN: one non-finite base value would poison all rows.  So the product runs on
    a NaN/inf-free copy of the base vector and the rows that really use a non-finite value are set to NaN afterwards (a second
    mat-vec with the 0/1 usage pattern); check flags travel as ``passthrough`` and never enter the product at all."""
    items = [LinearLoss.of(x) for x in items]
    if not items:
        return torch.stack([t.detach().reshape(()).float() for t in passthrough])
    base: Dict[int, Tuple[int, Tensor]] = {}
    offset = 0
    for it in items:
        for t, _ in it.terms:
            if id(t) not in base:
                base[id(t)] = (offset, t)
                offset += t.numel()
    tensors = [t for _, t in base.values()]
    dev = tensors[0].device
    with torch.no_grad():
        flat = torch.cat([t.detach().reshape(-1).float() for t in tensors]) if len(tensors) > 1 else tensors[0].detach().reshape(-1).float()
        key = ("m", str(dev), tuple((tuple((base[id(t)][0], t.numel(), c) for t, c in it.terms)) for it in items), offset)

        def build():
            m = torch.zeros(len(items), offset, dtype=torch.float32)
            for r, it in enumerate(items):
                for t, c in it.terms:
                    o = base[id(t)][0]
                    m[r, o:o + t.numel()] += c
            return m.to(dev), (m != 0).float().to(dev)
        coeff, uses = _const(key, build)
        bad = ~torch.isfinite(flat)
        out = coeff @ torch.where(bad, torch.zeros_like(flat), flat)
        out = torch.where((uses @ bad.float()) > 0, torch.full_like(out, float("nan")), out)
        if passthrough:
            out = torch.cat([out] + [t.detach().reshape(1).float() for t in passthrough])
        return out
