"""One training iteration replayed from the library's launch tape (include/miseg_hip.h, "Launch tape"; csrc/tape.hip).

The iteration of ``semi_seg.epocher.TrainEpocher`` (ref semi_seg/epocher.py:143-187) is ~300 launches issued from Python on two
threads; with fixed shapes they are the same (entry point, arguments, stream) list every time.  ``StepTape`` lets a few iterations run
eagerly (allocator pools, cached constants, packed-weight registry, seed registrations all settle), then runs ONE more eager iteration
with the library recording its own calls -- that iteration is an ordinary training step --, and from then on issues the recorded list
with one C call per iteration (``miseg_tape_replay``).  Per replayed iteration the host only draws the flip decisions, stages the
iteration's scalars in the pinned block (``StepIO.stage``), replays and reads the previous iteration's values.

What makes the recorded pointers stay valid:
  * every allocation of the recorded iteration (all threads, all streams) comes from a private ``torch.cuda.MemPool`` that nothing
    allocates from afterwards, and tensors handed between streams (``record_stream``) are kept alive until the iteration ends, so no
    block of the pool is reused across streams inside the recorded iteration on the strength of a host-side observation;
  * parameters, optimiser state, BatchNorm buffers, packed weights, the ``StepIO`` blocks are persistent;
  * the loader's batch tensors, the pinned staging slots and the events change per iteration: they are bound
    (``miseg_tape_bind``) and re-based at every replay.
What makes the replay complete: while recording, a ``TorchDispatchMode`` watches every ATen operator of both threads; an operator that
would launch device work outside the library (a fill, a copy, an add of two gradients ...) voids the recording -- the iteration has
run eagerly anyway -- and the epocher stays eager (it says why once).  The shipped trainers record clean (tests/test_gpu_step.py).
"""
from __future__ import annotations

import ctypes
import os
import warnings
from typing import List, Optional

import torch
from torch import Tensor
from torch.utils._python_dispatch import TorchDispatchMode

from . import _cabi, stepio

# ATen operators that launch nothing on the device: metadata, views, allocation.
_HARMLESS = {
    "aten::empty.memory_format", "aten::empty_strided", "aten::empty_like", "aten::new_empty", "aten::new_empty_strided",
    "aten::view", "aten::_unsafe_view", "aten::as_strided", "aten::narrow", "aten::slice.Tensor", "aten::select.int", "aten::detach",
    "aten::alias", "aten::t", "aten::transpose.int", "aten::permute", "aten::squeeze", "aten::squeeze.dim", "aten::squeeze.dims",
    "aten::unsqueeze", "aten::expand", "aten::_reshape_alias", "aten::unbind.int", "aten::split.Tensor", "aten::split_with_sizes",
    "aten::unsafe_split.Tensor", "aten::record_stream", "aten::is_pinned", "aten::lift_fresh", "aten::view.dtype",
    "aten::is_same_size", "aten::result_type.Tensor", "aten::unfold", "aten::diagonal", "aten::chunk", "aten::reshape", "aten::flatten.using_ints",
    "aten::sym_size.int", "aten::sym_stride.int", "aten::sym_numel", "aten::sym_storage_offset", "aten::is_contiguous",
    "aten::is_contiguous.memory_format", "aten::stride.int", "aten::size.int", "aten::dim", "aten::numel", "aten::_local_scalar_dense_cpu",
    "prim::device", "prim::dtype", "prim::layout", "aten::is_non_overlapping_and_dense", "aten::sym_is_contiguous",
}


def _has_cuda(x) -> bool:
    if isinstance(x, Tensor):
        return x.is_cuda
    if isinstance(x, (list, tuple)):
        return any(_has_cuda(v) for v in x)
    return False


class ForeignOps(TorchDispatchMode):
    """Collects the ATen operators that touch device tensors and are not known to be launch-free."""

    def __init__(self):
        super().__init__()
        self.seen: List[str] = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        if _IN_HOST_CALL:
            return out
        name = func._schema.name + ("." + func._schema.overload_name if func._schema.overload_name else "")
        if name not in _HARMLESS and (_has_cuda(args) or _has_cuda(list(kwargs.values())) or _has_cuda(out)):
            import traceback
            where = [f"{os.path.basename(fr.filename)}:{fr.lineno}" for fr in traceback.extract_stack(limit=12)
                     if "site-packages" not in fr.filename and "dist-packages" not in fr.filename and not fr.filename.endswith("tape.py")]
            try:
                node = torch._C._current_autograd_node()
            except Exception:
                node = None
            shape = tuple(out.shape) if isinstance(out, Tensor) else ""
            self.seen.append(name + (f"{shape}" if shape != "" else "") + (" in " + node.name() if node is not None else "") +
                             (" @ " + " < ".join(reversed(where[-3:])) if where else ""))
        return out


KEEP_ALIVE: Optional[list] = None      # while recording: tensors handed to another stream (see keep)
HOST_CALLS: Optional[list] = None      # while recording: the host-side calls between tape segments (see host_call)
_IN_HOST_CALL = 0


def host_call(fn):
    """Run ``fn()`` -- host-side work in the MIDDLE of an iteration that issues device work the library does not see (an RCCL
    all-reduce through torch.distributed, the wait for it).  While an iteration is being recorded the call also cuts the tape: the
    launches so far form one segment, ``fn`` is remembered, and a replay runs segment, ``fn()``, next segment ... in the recorded
    order.  ``fn`` must not depend on the thread that calls it (it sets its own stream): at replay that is the caller's thread, at
    record time it may be autograd's.  What ``fn`` launches is its own business -- the recording's ATen watch ignores it."""
    global _IN_HOST_CALL
    if _cabi.RECORDING and HOST_CALLS is not None:
        _cabi.lib().miseg_tape_mark(_cabi.RECORDING)
        HOST_CALLS.append(fn)
    _IN_HOST_CALL += 1
    try:
        return fn()
    finally:
        _IN_HOST_CALL -= 1


def keep(t: Optional[Tensor], stream) -> None:
    """``t.record_stream(stream)`` -- and, while an iteration is being recorded, a reference that outlives the iteration: the allocator
    may otherwise hand the block to another stream as soon as IT has seen the side stream's work finish, which a replay cannot know."""
    if t is None or not t.is_cuda:
        return
    t.record_stream(stream)
    if KEEP_ALIVE is not None:
        KEEP_ALIVE.append(t)


class StepTape:
    """Drives ``epocher._run_step(io, li, lt, ui, flip_masks)`` eagerly, then recorded once, then replayed."""

    MAX_ATTEMPTS = 3

    def __init__(self, epocher, warmup: int = 3):
        self.ep = epocher
        self.warmup = max(int(warmup), 2)
        self.seen = 0
        self.handle = 0
        self.key = None
        self.attempts = 0
        self.disabled: Optional[str] = None
        self.pool = None
        self.static = None          # (fields, names, items, nvals) of the recorded iteration's ticket
        self.bind_order: List[str] = []
        self.n_ops = 0
        self.replays = 0
        self.tags: list = []        # (op index, tag, (flops, bytes)) of the recorded iteration's tagged calls
        self.host_calls: list = []  # host-side calls between the tape's segments, in order (host_call)
        self._values = None

    def __del__(self):
        self.release()

    def release(self) -> None:
        if self.handle:
            try:
                _cabi.lib().miseg_tape_free(self.handle)
            except Exception:
                pass
            self.handle = 0
        self.pool = None

    # ------------------------------------------------------------------ per iteration
    def _key(self, io, li: Tensor, lt: Tensor, ui: Tensor, n_masks: int):
        from . import ops, unet_ops
        return (tuple(li.shape), tuple(lt.shape), tuple(ui.shape), li.dtype, lt.dtype, ui.dtype, li.is_contiguous(), lt.is_contiguous(),
                ui.is_contiguous(), n_masks, ops._stream(), ops._mi_precision, id(io), unet_ops.PACK_CACHE.generation,
                self.ep._model.training, self.ep._tape_signature(), id(getattr(self.ep, "_reducer", None)))

    def step(self, io, li: Tensor, lt: Tensor, ui: Tensor, flip_masks):
        ep = self.ep
        if self.disabled is not None:
            return ep._run_step(io, li, lt, ui, flip_masks)
        key = self._key(io, li, lt, ui, len(flip_masks))
        if self.handle and key == self.key:
            return self._replay(io, li, lt, ui, flip_masks)
        if self.handle:
            if key[-3:] != self.key[-3:]:   # the persistent state moved or the trainer changed: the recorded pointers are void
                self.release()
                self.seen = 0
            return ep._run_step(io, li, lt, ui, flip_masks)      # (another batch shape, e.g. a short last batch: eager, the tape stays)
        if self.seen < self.warmup or not (li.is_contiguous() and lt.is_contiguous() and ui.is_contiguous()):
            self.seen += 1
            return ep._run_step(io, li, lt, ui, flip_masks)
        return self._record(io, li, lt, ui, flip_masks, key)

    # ------------------------------------------------------------------ record
    def _record(self, io, li, lt, ui, flip_masks, key):
        global KEEP_ALIVE, HOST_CALLS
        ep = self.ep
        lib = _cabi.lib()
        dev = li.device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        pool = torch.cuda.MemPool()
        guard = ForeignOps()
        KEEP_ALIVE, HOST_CALLS = [], []
        handle = lib.miseg_tape_begin()
        if not handle:
            KEEP_ALIVE = HOST_CALLS = None
            return ep._run_step(io, li, lt, ui, flip_masks)
        torch._C._cuda_beginAllocateToPool(idx, pool.id)       # every thread's allocations (the backward pass runs on the engine's)
        _cabi.RECORDING, _cabi.TAPE_TAGS = handle, []
        ticket = None
        try:
            with guard:
                ticket = ep._run_step(io, li, lt, ui, flip_masks)
        finally:
            torch._C._cuda_endAllocateToPool(idx, pool.id)
            torch._C._cuda_releasePool(idx, pool.id)
            lib.miseg_tape_end(handle)
            _cabi.RECORDING = 0
            host_calls, HOST_CALLS = HOST_CALLS, None
            KEEP_ALIVE = None
        self.tags = list(_cabi.TAPE_TAGS)
        self.attempts += 1
        if guard.seen or not isinstance(ticket, stepio.Ticket):
            lib.miseg_tape_free(handle)
            why = ("device work outside the library: " + ", ".join(sorted(set(guard.seen)))) if guard.seen else "the iteration posted no StepIO ticket"
            if self.attempts >= self.MAX_ATTEMPTS:
                self.disabled = why
                if os.environ.get("MISEG_TAPE_QUIET", "0") != "1":
                    warnings.warn(f"launch tape not used, the iteration stays eager ({why})")
            return ticket
        # bindings, in the order _replay passes their values
        self.bind_order = []

        def bind(name, ptr, span):
            slot = lib.miseg_tape_bind(handle, ptr, span)
            assert slot == len(self.bind_order), (name, slot)
            self.bind_order.append(name)
            return lib.miseg_tape_bind_uses(handle, slot)
        uses = {
            "li": bind("li", li.data_ptr(), li.numel() * li.element_size()),
            "lt": bind("lt", lt.data_ptr(), lt.numel() * lt.element_size()),
            "ui": bind("ui", ui.data_ptr(), ui.numel() * ui.element_size()),
            "up": bind("up", io.host[io.turn].data_ptr(), stepio.PARAM_BYTES),
            "up_ev": bind("up_ev", io.up_events[io.turn], 1),
            "out": bind("out", io.out_host[ticket.slot].data_ptr(), stepio.OUT_BYTES),
            "out_ev": bind("out_ev", io.out_events[ticket.slot], 1),
        }
        if uses["up"] != 1 or uses["up_ev"] != 1 or uses["out"] != 1 or uses["out_ev"] != 1:
            lib.miseg_tape_free(handle)
            self.disabled = f"unexpected staging traffic in the recorded iteration ({uses})"
            warnings.warn(f"launch tape not used, the iteration stays eager ({self.disabled})")
            return ticket
        if lib.miseg_tape_segments(handle) != len(host_calls) + 1:
            lib.miseg_tape_free(handle)
            self.disabled = "segment marks and host calls of the recorded iteration do not pair up"
            warnings.warn(f"launch tape not used, the iteration stays eager ({self.disabled})")
            return ticket
        self.handle, self.key, self.pool, self.host_calls = handle, key, pool, host_calls
        self.static = (ticket.fields, ticket.names, ticket.items, ticket.nvals)
        self.n_ops = lib.miseg_tape_len(handle)
        self.uses = uses
        self._values = (ctypes.c_void_p * len(self.bind_order))()
        return ticket

    # ------------------------------------------------------------------ replay
    def _replay(self, io, li, lt, ui, flip_masks):
        ep = self.ep
        lib = _cabi.lib()
        slot = ep._stage(io, flip_masks)                     # host half: flips, Adam's scalars, seeds -> pinned slot
        io.out_turn = (io.out_turn + 1) % len(io.out_events)
        oslot = io.out_turn
        v = self._values
        v[0], v[1], v[2] = li.data_ptr(), lt.data_ptr(), ui.data_ptr()
        v[3], v[4] = io.host[slot].data_ptr(), io.up_events[slot]
        v[5], v[6] = io.out_host[oslot].data_ptr(), io.out_events[oslot]
        for seg in range(len(self.host_calls) + 1):
            rc = lib.miseg_tape_replay(self.handle, seg, v, len(v))
            if rc != 0:
                msg = lib.miseg_last_error()
                raise _cabi.MisegError(f"miseg_tape_replay failed ({rc}): {msg.decode() if msg else ''}")
            if seg < len(self.host_calls):
                self.host_calls[seg]()         # e.g. the all-reduce of a gradient bucket (miseg_amd.ddp.GradReducer)
        io.up_pending[slot] = True
        self.replays += 1
        ep._after_replay()
        fields, names, items, nvals = self.static
        return stepio.Ticket(io, oslot, fields, names, items, nvals)

    # ------------------------------------------------------------------ introspection (bench.py)
    def op_names(self) -> List[str]:
        lib = _cabi.lib()
        return [lib.miseg_tape_op_name(self.handle, i).decode() for i in range(self.n_ops)] if self.handle else []

    def time_tag(self, tag: str) -> List[int]:
        """HIP-event pairs around every op of the recorded iteration tagged ``tag``, at every replay from now on."""
        lib = _cabi.lib()
        ops_ = [i for i, t, _ in self.tags if t == tag]
        for i in ops_:
            lib.miseg_tape_time_op(self.handle, i)
        return ops_

    def timed_ms(self, op: int, cap: int = 4096) -> List[float]:
        buf = (ctypes.c_float * cap)()
        n = _cabi.lib().miseg_tape_timed_ms(self.handle, op, buf, cap)
        return [buf[i] for i in range(max(n, 0))]
