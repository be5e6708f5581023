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

    def backward(self, scale: float = 1.0) -> None:
        """Seed autograd at the kernel outputs with constant gradients ``coefficient * scale`` (``scale`` = the loss scale of the fp16
        storage mode).  Inside an iteration the constants live in the step block (``stepio.StepIO.seed``: refreshed by the iteration's
        one upload with the CURRENT scale, so a dynamic loss scale needs no new tensors); otherwise cached ``torch.full`` tensors."""
        from . import stepio
        io = stepio.CURRENT
        roots, grads = [], []
        for t, c in self.merged():
            if t.requires_grad:
                roots.append(t)
                if io is not None and t.is_cuda and t.dtype == torch.float32 and io.device == t.device and io.scale == float(scale):
                    grads.append(io.seed(t.shape, c))
                else:
                    grads.append(_const(("g", tuple(t.shape), t.dtype, str(t.device), c * scale),
                                        lambda t=t, c=c: torch.full(t.shape, c * scale, dtype=t.dtype, device=t.device)))
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


_FUSED_REPORT = True     # False: the torch-op evaluation on the GPU too (a dozen one-element launches; tests compare the two)


def report(items: Sequence[Union[LinearLoss, Tensor]], check_items: Sequence[tuple] = ()) -> Tensor:
    """float32 device vector: the value of every item, then the flag of every deferred check (miseg_amd.checks item tuples), in
    order -- what ``evaluate(items, passthrough=[flag_tensor(c) ...])`` returns, but on the GPU as (at most) two ``torch.cat`` and
    ONE launch of ``miseg_report_scalars`` instead of ~20 one-element kernels (isfinite / where / two mat-vecs; isnan-sum-cast per
    NaN test; a cast per integer flag).  Falls back to ``evaluate`` off the GPU or for flag kinds the kernel does not know."""
    from . import checks as _checks
    from ._cabi import call
    items = [LinearLoss.of(x) for x in items]
    srcs = [c[0] for c in check_items]
    tensors_all = [t for it in items for t, _ in it.terms] + [s.tensor if isinstance(s, _checks.LazyFlag) else s for s in srcs if not callable(s) or isinstance(s, _checks.LazyFlag)]
    fused = bool(tensors_all) and all(torch.is_tensor(t) and t.is_cuda for t in tensors_all) and \
        all(isinstance(s, _checks.LazyFlag) or (torch.is_tensor(s) and s.dtype == torch.float32) for s in srcs) and \
        all(s.tensor.dtype == (torch.float32 if s.kind == "nan" else torch.int32) for s in srcs if isinstance(s, _checks.LazyFlag))
    if not fused or not _FUSED_REPORT:
        return evaluate(items, passthrough=[_checks.flag_tensor(c) for c in check_items])
    dev = tensors_all[0].device
    cur = torch.cuda.current_stream(dev)
    from . import stepio
    io = stepio.CURRENT
    if io is not None and io.device == dev:
        out = _report_in_place(io, items, srcs, _checks, cur)
        if out is not None:
            return out
    fl: List[Tensor] = []          # float tensors, in flat order
    off: Dict[int, int] = {}
    n_f = 0

    def add_f(t: Tensor) -> int:
        nonlocal n_f
        o = off.get(id(t))
        if o is None:
            o = off[id(t)] = n_f
            fl.append(t)
            n_f += t.numel()
        return o
    for it in items:
        for t, _ in it.terms:
            add_f(t)
    C = n_f
    il: List[Tensor] = []
    desc: List[int] = []
    for s in srcs:
        if isinstance(s, _checks.LazyFlag) and s.kind == "cast":
            desc += [2, len(il), 0]
            il.append(s.tensor)
        elif isinstance(s, _checks.LazyFlag):
            desc += [1, add_f(s.tensor), s.tensor.numel()]
        else:
            desc += [0, add_f(s), 1]
    for t in fl + il:
        t.record_stream(cur)          # some were produced on the IIC branch stream
    with torch.no_grad():
        flat = torch.cat([t.detach().reshape(-1).float() for t in fl]) if len(fl) > 1 else fl[0].detach().reshape(-1).float()
        iflat = None if not il else (torch.cat([t.detach().reshape(-1) for t in il]) if len(il) > 1 else il[0].detach().reshape(-1))
        key = ("rep", str(dev), tuple(tuple((off[id(t)], t.numel(), c) for t, c in it.terms) for it in items), C, tuple(desc))

        def build():
            m = torch.zeros(max(len(items), 1), max(C, 1), dtype=torch.float32)
            for r, it in enumerate(items):
                for t, c in it.terms:
                    o = off[id(t)]
                    m[r, o:o + t.numel()] += c
            d = torch.tensor(desc if desc else [0, 0, 0], dtype=torch.int32)
            return m.to(dev), d.to(dev)
        coeff, d_dev = _const(key, build)
        out = torch.empty(len(items) + len(srcs), dtype=torch.float32, device=dev)
        call("miseg_report_scalars", cur.cuda_stream, flat.data_ptr(), None if iflat is None else iflat.data_ptr(), coeff.data_ptr(),
             len(items), C, d_dev.data_ptr(), len(srcs), out.data_ptr())
    return out


def _report_in_place(io, items, srcs, _checks, cur) -> "Tensor | None":
    """``report`` when every value lives in the iteration's scalar arena and every integer flag in its counter block
    (``stepio.StepIO.scalar`` / ``counter``: the loss kernels and the heads wrote them there): the kernel reads the arena in place
    -- no ``torch.cat`` -- and writes into the iteration's read-back block.  One slot is left free behind the flags: the fp16 mode's
    overflow count goes there, so that the optimiser's guard stays one contiguous vector.  None if anything lives elsewhere."""
    from ._cabi import call
    cols: List[Tuple[int, int]] = []
    for it in items:
        for t, _ in it.terms:
            o = io.arena_offset(t)
            if o is None:
                return None
            cols.append((o, t.numel()))
    desc: List[int] = []
    for s in srcs:
        if isinstance(s, _checks.LazyFlag) and s.kind == "cast":
            i = io.counter_index(s.tensor)
            if i is None or s.tensor.numel() != 1:
                return None
            desc += [2, i, 0]
        elif isinstance(s, _checks.LazyFlag):
            o = io.arena_offset(s.tensor)
            if o is None:
                return None
            desc += [1, o, s.tensor.numel()]
        else:
            o = io.arena_offset(s)
            if o is None or s.numel() != 1:
                return None
            desc += [0, o, 1]
    C = io._cursor_arena
    dev = io.device
    key = ("rep_io", str(dev), tuple(tuple((io.arena_offset(t), t.numel(), c) for t, c in it.terms) for it in items), C, tuple(desc))

    def build():
        m = torch.zeros(max(len(items), 1), max(C, 1), dtype=torch.float32)
        for r, it in enumerate(items):
            for t, c in it.terms:
                o = io.arena_offset(t)
                m[r, o:o + t.numel()] += c
        d = torch.tensor(desc if desc else [0, 0, 0], dtype=torch.int32)
        return m.to(dev), d.to(dev)
    coeff, d_dev = _const(key, build)
    n = len(items) + len(srcs)
    out = io.out("scalars", (n + 1,), torch.float32)
    io.last_report = out
    call("miseg_report_scalars", cur.cuda_stream, io.arena.data_ptr(), io.counters_base().data_ptr(), coeff.data_ptr(), len(items), C,
         d_dev.data_ptr(), len(srcs), out.data_ptr())
    return out[:n]
