"""Device-side assertions that do not stall the stream.

The reference asserts ``simplex(prob)`` and raises on NaN losses inline (contrastyou/losses/iic_loss.py:28-29,
102-104,132-133,184-186; whl:deepclustering2/loss/kl_losses.py), each of which is a blocking device->host read.
Here every check produces a 0-d device tensor (non-zero = failed).  Outside a ``deferred`` block the check is
evaluated at once (same behaviour as the reference).  Inside one -- the train epocher opens it around an iteration --
the flags ride along with the iteration's single host copy and the same exception is raised there.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
from torch import Tensor

from ._cabi import call

_Item = Tuple[object, Callable[[str], BaseException], str]   # (0-d flag tensor | () -> flag tensor, exception type, message)
_active: Optional[List[_Item]] = None


class deferred:
    """``with deferred(items):`` -- checks issued inside are appended to ``items`` instead of being evaluated."""

    def __init__(self, items: List[_Item]):
        self.items = items

    def __enter__(self):
        global _active
        self._prev, _active = _active, self.items
        return self

    def __exit__(self, *exc):
        global _active
        _active = self._prev
        return False


class LazyFlag:
    """A check whose 0-d float flag has not been computed yet: ``kind`` 'nan' (number of NaNs in ``tensor``) or 'cast' (``tensor`` is
    a 0-d count of another dtype).  Calling it computes the flag with torch ops (a few one-element launches); the iteration's
    fused report (miseg_amd.lazy.report -> miseg_report_scalars) reads ``tensor`` directly instead."""
    __slots__ = ("kind", "tensor")

    def __init__(self, kind: str, tensor: Tensor):
        self.kind, self.tensor = kind, tensor

    def __call__(self) -> Tensor:
        t = self.tensor
        if t.is_cuda:
            t.record_stream(torch.cuda.current_stream(t.device))   # produced on the branch stream, read here
        return torch.isnan(t).sum().reshape(()).float() if self.kind == "nan" else t.reshape(()).float()


def require_zero(bad: Tensor, exc, msg: str) -> None:
    """Fail with ``exc(msg)`` if the 0-d device tensor ``bad`` is non-zero (now, or at the enclosing block's fetch)."""
    if _active is None:
        if float(bad) != 0.0:
            raise exc(msg)
    else:
        b = bad.detach()
        if b.dtype == torch.float32 and b.dim() == 0:
            _active.append((b, exc, msg))
        else:   # the cast is one more tiny launch: issued when the flags are gathered, not here (see raise_if_nan)
            _active.append((LazyFlag("cast", b), exc, msg))


def raise_failed(items: List[_Item], host_flags) -> None:
    for (_, exc, msg), flag in zip(items, host_flags):
        if flag != 0.0:
            raise exc(msg)


def simplex_violations(t: Tensor, axis: int = 1, tol: float = 2e-4) -> Tensor:
    """0-d int32 device count of positions whose sum over ``axis`` is not within tol of 1 (tol = allclose's
    atol + rtol*1 of the reference's ``simplex``)."""
    cached = getattr(t, "_miseg_simplex", None)   # counted by the producing kernel (ops.local_head), tensor unchanged since
    if cached is not None and cached[0] == axis % t.dim() and cached[2] == t._version and tol == 2e-4:
        return cached[1]
    if not t.is_cuda or t.dtype != torch.float32:
        s = t.float().sum(axis)
        return (~((s - 1.0).abs() <= tol)).sum().to(torch.int32)
    t = t.contiguous()
    axis = axis % t.dim()
    outer = 1
    for d in t.shape[:axis]:
        outer *= d
    inner = 1
    for d in t.shape[axis + 1:]:
        inner *= d
    from .ops import zero_counter
    count = zero_counter(t.device)         # inside an iteration: a counter of the step block (zeroed by its upload), else torch.zeros
    if t.numel():
        call("miseg_simplex_violations", torch.cuda.current_stream().cuda_stream, t.data_ptr(), outer, t.shape[axis], inner,
             float(tol), count.data_ptr())
    return count


def assert_simplex(t: Tensor, axis: int = 1, msg: str = "") -> None:
    require_zero(simplex_violations(t, axis), AssertionError, msg)


def raise_if_nan(values: Tensor, describe: str = "nan loss") -> None:
    """ref iic_loss.py:132-133 / :184-186 raise RuntimeError when a (patch) loss is NaN.  Inside a ``deferred`` block the
    three small kernels of the test (isnan, sum, cast) are not launched here -- in line they sit on the IIC branch's stream
    between the loss and its backward, i.e. on the step's critical path -- but when the block's flags are gathered
    (``flag_tensor``), on the stream that makes the iteration's host copy."""
    if _active is None:
        require_zero(torch.isnan(values).sum(), RuntimeError, describe)
        return
    _active.append((LazyFlag("nan", values.detach()), RuntimeError, describe))


def flag_tensor(item: _Item) -> Tensor:
    """The 0-d float flag of a deferred item (evaluates the lazily recorded ones)."""
    return item[0]() if callable(item[0]) else item[0]
