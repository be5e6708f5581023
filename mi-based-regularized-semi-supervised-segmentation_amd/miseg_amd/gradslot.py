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
