"""Hand-out of flat-gradient slots to the backward of our own autograd nodes (see miseg_amd.flat.FlatBuffers)."""
from __future__ import annotations

from typing import Optional

from torch import Tensor


def grad_slot(param) -> Optional[Tensor]:
    """For the backward of our own autograd nodes: a fresh view of ``param``'s slice of the flat gradient buffer if the
    kernel may write the gradient there directly (the parameter belongs to a FlatBuffers, has no gradient yet in this
    backward pass and nobody claimed the slot), else None (the caller allocates; autograd then accumulates as usual).
    Returning that view from ``backward`` lets AccumulateGrad adopt it as ``param.grad`` without launching a kernel."""
    slot = getattr(param, "_miseg_grad_slot", None)
    if slot is None or param.grad is not None or getattr(param, "_miseg_grad_claimed", True):
        return None
    param._miseg_grad_claimed = True
    return slot.detach()


# ------------------------------------------------------------------------------------------ adjacent parameter groups
# The cluster heads keep the reference's module tree (one Linear / Conv2d per sub-head), but their kernels want the S
# weights as one [S, ...] tensor.  Parameters registered here as a group are laid out back to back by FlatBuffers, so the
# stacked tensor and its gradient are VIEWS of the flat buffers: no torch.stack, no unbind copies, no AccumulateGrad adds.
import weakref as _weakref
from typing import List as _List, Sequence as _Sequence

import torch as _torch

_GROUPS: _List[_List["_weakref.ReferenceType"]] = []


def register_adjacent(params: _Sequence[_torch.nn.Parameter]) -> None:
    _GROUPS.append([_weakref.ref(p) for p in params])


def adjacent_groups() -> _List[_List[_torch.nn.Parameter]]:
    out = []
    for g in _GROUPS:
        ps = [r() for r in g]
        if all(p is not None for p in ps):
            out.append(ps)
    return out


def _adjacent(tensors) -> bool:
    t0 = tensors[0]
    if not t0.is_contiguous():
        return False
    step = t0.numel() * t0.element_size()
    base = t0.untyped_storage().data_ptr()      # back to back in ONE storage (the flat buffer): separate allocations that merely happen to be neighbours do not count
    return all(t.shape == t0.shape and t.dtype == t0.dtype and t.is_contiguous() and t.data_ptr() == t0.data_ptr() + i * step
               and t.untyped_storage().data_ptr() == base for i, t in enumerate(tensors))


def _stack_view(t0: Tensor, n: int) -> Tensor:
    return t0.as_strided((n,) + tuple(t0.shape), (t0.numel(),) + tuple(t0.stride()))


def stacked_grad_slot(params) -> Optional[Tensor]:
    """One [S, ...] view over the flat-gradient slots of ``params`` if they are back to back and all still open."""
    if not params:
        return None
    slots = [getattr(p, "_miseg_grad_slot", None) for p in params]
    if any(s is None for s in slots) or any(p.grad is not None or getattr(p, "_miseg_grad_claimed", True) for p in params):
        return None
    if not _adjacent(slots):
        return None
    for p in params:
        p._miseg_grad_claimed = True
    return _stack_view(slots[0].detach(), len(slots))


class _AdjacentStack(_torch.autograd.Function):
    @staticmethod
    def forward(ctx, *params):
        ctx.n = len(params)
        return _stack_view(params[0].detach(), len(params))

    @staticmethod
    def backward(ctx, g):
        return tuple(g[i] for i in range(ctx.n))     # views: adopted as .grad without a kernel when g lives in the flat buffer


def stacked_param(params) -> Tensor:
    """torch.stack(params) -- as a zero-copy view when the parameters are adjacent in memory (see register_adjacent)."""
    params = list(params)
    if len(params) > 1 and _adjacent([p.data for p in params]):
        out = _AdjacentStack.apply(*params)
        out._miseg_stack_params = params
        return out
    return _torch.stack(params)
