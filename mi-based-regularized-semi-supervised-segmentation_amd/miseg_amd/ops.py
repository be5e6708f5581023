"""torch.autograd front-ends of the HIP kernels (thin: pointer/shape plumbing only).

Every function here requires CUDA(HIP) tensors and raises otherwise -- the product has no CPU or
eager-PyTorch fallback.  PyTorch supplies device memory, the current HIP stream and autograd
bookkeeping; all arithmetic happens in ``csrc/*.hip`` behind ``include/miseg_hip.h``.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _cabi, stepio
from ._cabi import BF16, F16, F32, call, query
from .tape import keep

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """The current HIP stream's handle.  Asked ~90 times per training step: `torch.cuda.current_stream().cuda_stream` builds a Python
    Stream object each time (9 us, 0.85 ms of host time per step -- the host is co-critical at this step time); the two raw
    bindings cost well under a microsecond."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts: Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _cabi.MisegError("miseg_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_LIB_FORK = True         # False: torch.cuda.Stream.wait_stream (system-scope events)


def wait_stream(waiter: "torch.cuda.Stream", producer: "torch.cuda.Stream") -> None:
    """``waiter.wait_stream(producer)`` through the library's device-scope-release events (``miseg_stream_wait_stream``): a torch event
    releases to system scope, which costs the PRODUCING stream ~5 us before its next kernel (scratch/fork_cost.py: 90.6 -> 88.2 us per
    kernel pair with a fork in between, 85.4 without) -- 22 weight-gradient forks per backward pass.  Under stream capture the torch
    call stays (the capture has to see the dependency)."""
    if waiter == producer:
        return
    if not _LIB_FORK or torch.cuda.is_current_stream_capturing():
        waiter.wait_stream(producer)
        return
    call("miseg_stream_wait_stream", waiter.cuda_stream, producer.cuda_stream)


def scalar_out(shape, device) -> Tensor:
    """fp32 tensor for a kernel's small result: inside an iteration it lives in the iteration's scalar arena (stepio), where the
    report kernel reads it in place; otherwise an ordinary allocation."""
    io = stepio.CURRENT
    if io is not None and io.device == device:
        return io.scalar(shape)
    return torch.empty(tuple(shape), dtype=torch.float32, device=device)


def zero_counter(device, shape=()) -> Tensor:
    """int32 counter that is zero before the kernel that adds to it runs: inside an iteration the step block's upload has zeroed it
    (no fill launch); otherwise torch.zeros."""
    io = stepio.CURRENT
    if io is not None and io.device == device:
        return io.counter().view(tuple(shape))
    return torch.zeros(tuple(shape), dtype=torch.int32, device=device)


def fill_zero(t: Tensor) -> Tensor:
    """``t.zero_()`` as a library launch (dense tensors; it is on the launch tape)."""
    assert t.is_cuda and (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last))
    call("miseg_fill_zero", _stream(), t.data_ptr(), t.numel() * t.element_size())
    return t


def _ws(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


_CONST_CACHE = {}


def cached_const(key, build):
    """Small static device tensors (window lists, gather indices) are built once per (shape, device)."""
    t = _CONST_CACHE.get(key)
    if t is None:
        t = _CONST_CACHE[key] = build()
    return t


def windows_tensor(windows, device) -> Tensor:
    key = ("win", str(device), tuple(tuple(int(v) for v in w) for w in windows))
    return cached_const(key, lambda: torch.tensor([list(w) for w in windows], dtype=torch.int32, device=device).view(len(windows), 4))


def arange_i32(start: int, stop: int, device) -> Tensor:
    """Cached int32 [start, stop) on the device; carries its bounds on the host (``_miseg_range``) so that a consumer that
    scatters into rows ``src`` (local_head's backward) knows which rows it will NOT write without reading the tensor back."""
    def build():
        t = torch.arange(start, stop, dtype=torch.int32, device=device)
        t._miseg_range = (int(start), int(stop))
        return t
    return cached_const(("arange", str(device), start, stop), build)


def flip_masks(decisions: Sequence[Sequence[bool]]) -> List[int]:
    """[[flip_h, flip_w], ...] -> bit masks (bit0 = H, bit1 = W), still on the host."""
    return [int(bool(fh)) | (int(bool(fw)) << 1) for fh, fw in decisions]


def flips_to_tensor(decisions: Sequence[Sequence[bool]], device) -> Tensor:
    """[[flip_h, flip_w], ...] -> int32 bit masks (bit0 = H, bit1 = W)."""
    return torch.tensor(flip_masks(decisions), dtype=torch.int32, device=device)



class PinnedRing:
    """``slots`` pinned host buffers of one shape used round-robin as the source of non-blocking host->device copies.  Every slot
    remembers an event recorded behind its last copy and waits for it before the slot is rewritten: free while the host is fewer than
    ``slots`` copies ahead of the copy engine, and a loop WITHOUT a per-iteration host synchronisation (a user loop, `bench.py
    --workload input`) can no longer overwrite a slot whose copy is still queued."""

    def __init__(self, shape, dtype, slots: int = 4):
        self.host = torch.empty((slots,) + tuple(shape), dtype=dtype).pin_memory()
        self.events = [None] * slots
        self.turn = 0

    def upload(self, fill, device) -> Tensor:
        """``fill(host_slot)`` writes the payload; returns a fresh device tensor holding it (copy enqueued on the current stream)."""
        self.turn = (self.turn + 1) % len(self.events)
        ev = self.events[self.turn]
        if ev is not None:
            ev.synchronize()
        slot = self.host[self.turn]
        fill(slot)
        out = torch.empty(slot.shape, dtype=slot.dtype, device=device)
        out.copy_(slot, non_blocking=True)
        ev = self.events[self.turn] = ev or torch.cuda.Event()
        ev.record()
        return out

    def upload_into(self, fill, dst: Tensor) -> None:
        """The same into an existing device tensor (static buffers of a captured step)."""
        self.turn = (self.turn + 1) % len(self.events)
        ev = self.events[self.turn]
        if ev is not None:
            ev.synchronize()
        slot = self.host[self.turn]
        fill(slot)
        dst.copy_(slot, non_blocking=True)
        ev = self.events[self.turn] = ev or torch.cuda.Event()
        ev.record()

# ------------------------------------------------------------------------------------------ local MI
MI_PRECISIONS = {"fp32": 0, "bf16x3": 1, "bf16": 2, "f16f8": 3}
_mi_precision = MI_PRECISIONS.get(__import__("os").environ.get("MISEG_MI_PRECISION", "fp32"), 0)


def set_mi_precision(mode: str) -> None:
    """Arithmetic of the local-MI contraction: 'fp32' (exact fp32 MFMA, default), 'bf16x3' (bf16 MFMA on hi/lo-split
    operands: fp32-class accuracy at 3/16 of the MFMA time), 'f16f8' (f16 hi x hi + both cross terms K-concatenated on the
    block-scaled fp8 pipe: the same accuracy class at 2/3 of bf16x3's matrix time; kernels that have no such form take bf16x3)
    or 'bf16' (plain bf16 operands)."""
    global _mi_precision
    _mi_precision = MI_PRECISIONS[mode]


def resolve_mi_precision(compute_dtype, requested: Optional[str] = None) -> str:
    """The local-MI arithmetic a trainer runs with: the ``MISEG_MI_PRECISION`` environment variable if set (an override for A/B
    runs), else the model's ``Arch.mi_precision`` config key, else by the U-Net's storage type -- exact fp32 MFMA for
    ``Arch.compute_dtype=float32`` (the parity mode), the f16 + fp8 split for bfloat16 / float16 (the benchmarked arithmetic)."""
    env = os.environ.get("MISEG_MI_PRECISION")
    mode = env or requested or ("fp32" if compute_dtype in (torch.float32, "float32", "fp32", None) else "f16f8")
    if mode not in MI_PRECISIONS:
        raise ValueError(f"mi_precision must be one of {sorted(MI_PRECISIONS)}, got {mode!r}")
    return mode


def mi_precision_name() -> str:
    return next(k for k, v in MI_PRECISIONS.items() if v == _mi_precision)


def colour_windows(windows: Sequence[Tuple[int, int, int, int]]) -> List[List[int]]:
    """Greedy colouring of overlapping windows into pairwise-disjoint groups (deterministic backward:
    each group is one launch with plain read-modify-write, groups run in stream order)."""
    groups: List[List[int]] = []
    for idx, (h0, h1, w0, w1) in enumerate(windows):
        for grp in groups:
            if all(h1 <= windows[j][0] or windows[j][1] <= h0 or w1 <= windows[j][2] or windows[j][3] <= w0 for j in grp):
                grp.append(idx)
                break
        else:
            groups.append([idx])
    return groups


def _local_fwd(x: Tensor, y: Tensor, mask, pad: int, windows, win: Tensor, raw: Tensor) -> None:
    """raw[P][T][T][K][K] of one (x, y) pair: both contiguous [N,K,H,W] fp32 (possibly views into a larger tensor)."""
    n, k, h, w = x.shape
    p, t = len(windows), 2 * pad + 1
    ws = _ws(query("miseg_iic_local_joint_ws_bytes", n, k, h, w, pad, p), x.device)
    px = sum((a1 - a0) * (b1 - b0) for a0, a1, b0, b1 in windows)
    call("miseg_iic_local_joint_fwd", _stream(), _ptr(x), _ptr(y), _ptr(mask), n, k, h, w, pad, _ptr(win), p, _ptr(raw),
         _ptr(ws), ws.numel(), _mi_precision, work=(2.0 * k * k * t * t * n * px, 2.0 * n * k * px * 4),
         tag=f"iic_local_joint_fwd[p{pad}]")


def _is_whole(windows, h: int, w: int) -> bool:
    return len(windows) == 1 and tuple(windows[0]) == (0, h, 0, w)


def _local_bwd(x: Tensor, y: Tensor, mask, pad: int, windows, win: Tensor, grad_raw: Tensor, scale: Tensor, gx: Tensor,
               gy: Tensor) -> None:
    """gx, gy (same layout as x, y): overwritten for a single whole-image window, accumulated into otherwise (caller
    zero-fills).  Overlapping windows are split into disjoint colour groups, one launch each, in stream order."""
    n, k, h, w = x.shape
    whole = _is_whole(windows, h, w)
    bws = _ws(query("miseg_iic_local_bwd_ws_bytes", k, pad, len(windows)), x.device)
    tt = (2 * pad + 1) ** 2
    for grp in colour_windows(windows):
        if len(grp) == len(windows):
            gwin, ggrad, gscale = win, grad_raw, scale
        else:
            idx = cached_const(("idx", str(x.device), tuple(grp)), lambda: torch.tensor(grp, dtype=torch.long, device=x.device))
            gwin, ggrad, gscale = win[idx].contiguous(), grad_raw[idx].contiguous(), scale[idx].contiguous()
        px = sum((windows[i][1] - windows[i][0]) * (windows[i][3] - windows[i][2]) for i in grp)
        call("miseg_iic_local_bwd", _stream(), _ptr(x), _ptr(y), _ptr(mask), n, k, h, w, pad, _ptr(gwin), len(grp),
             _ptr(ggrad), _ptr(gscale), _ptr(gx), _ptr(gy), 0 if whole else 1, _mi_precision, _ptr(bws), bws.numel(),
             work=(4.0 * k * k * tt * n * px, 4.0 * n * k * px * 4), tag=f"iic_local_bwd[p{pad}]")


class _LocalMI(torch.autograd.Function):
    """loss[P] of IIDSegmentationLoss over P windows (ref: contrastyou/losses/iic_loss.py:107-149)."""

    @staticmethod
    def forward(ctx, x: Tensor, y: Tensor, mask: Optional[Tensor], pad: int, windows, lamda: float):
        _need_gpu(x, y, mask)
        x, y = x.contiguous().float(), y.contiguous().float()
        mask = None if mask is None else mask.contiguous().float()
        n, k, h, w = x.shape
        p = len(windows)
        dev = x.device
        win = windows_tensor(windows, dev)
        t = 2 * pad + 1
        raw = torch.empty(p, t, t, k, k, dtype=torch.float32, device=dev)
        _local_fwd(x, y, mask, pad, windows, win, raw)
        loss = scalar_out((p,), dev)
        grad_raw = torch.empty_like(raw)
        _local_loss(raw, k, pad, p, lamda, loss, grad_raw)
        ctx.save_for_backward(x, y, mask, grad_raw, win)
        ctx.pad, ctx.windows = pad, list(windows)
        ctx.raw = raw
        return loss

    @staticmethod
    def backward(ctx, gloss: Tensor):
        x, y, mask, grad_raw, win = ctx.saved_tensors
        n, k, h, w = x.shape
        whole = _is_whole(ctx.windows, h, w)
        gx, gy = (torch.empty_like(x), torch.empty_like(y)) if whole else (torch.zeros_like(x), torch.zeros_like(y))
        _local_bwd(x, y, mask, ctx.pad, ctx.windows, win, grad_raw, gloss.contiguous().float(), gx, gy)
        return gx, gy, None, None, None, None


def _local_loss(raw: Tensor, k: int, pad: int, p: int, lamda: float, loss: Tensor, grad_raw: Tensor) -> None:
    """Loss epilogue of the local MI (ref iic_loss.py:124-146).  pad >= 2: one block per (window, displacement) -- the single
    block per window needs ceil((2 pad + 1)^2 / 16) rounds on one CU while the IIC chain waits for it."""
    if pad >= 2:
        ws = _ws(query("miseg_iic_local_loss_ws_bytes", pad, p), raw.device)
        call("miseg_iic_local_loss_fwd_ws", _stream(), _ptr(raw), k, pad, p, float(lamda), _ptr(loss), _ptr(grad_raw), _ptr(ws), ws.numel())
    else:
        call("miseg_iic_local_loss_fwd", _stream(), _ptr(raw), k, pad, p, float(lamda), _ptr(loss), _ptr(grad_raw))


# 1: the joint forward writes the backward's operand planes and the f16 + fp8 backward copies its source rows from them.  Built and
# measured: the backward gains less inside the step (-0.06 ms) than the by-product costs the forward (+0.07 ms).
_MI_PLANES = os.environ.get("MISEG_MI_PLANES", "0") == "1"


class _LocalMIHeads(torch.autograd.Function):
    """All S sub-heads of one decoder tap in one node: probs[S, 2*UB, K, H, W] holds, per sub-head, the UB maps of the
    flipped features followed by the UB maps of the transformed image (ref semi_seg/epocher.py:264-272 evaluates the
    criterion once per sub-head on ``p[:ub], p[ub:]``).  loss[S][P]; the backward writes every sub-head's gradient
    straight into one gprob[S, 2*UB, K, H, W] buffer, so autograd never materialises per-slice zero-padded copies."""

    @staticmethod
    def forward(ctx, probs: Tensor, ub: int, mask: Optional[Tensor], pad: int, windows, lamda: float):
        _need_gpu(probs, mask)
        probs = probs.contiguous().float()
        mask = None if mask is None else mask.contiguous().float()
        s, n2, k, h, w = probs.shape
        if n2 != 2 * ub:
            raise _cabi.MisegError(f"local_mi_heads: probs holds {n2} maps per sub-head, expected 2*{ub}")
        p, t, dev = len(windows), 2 * pad + 1, probs.device
        win = windows_tensor(windows, dev)
        raw = torch.empty(s, p, t, t, k, k, dtype=torch.float32, device=dev)
        planes = None
        if mask is None:    # every sub-head in one launch
            ws = _ws(max(query("miseg_iic_local_joint_ws_bytes", ub, k, h, w, pad, p * s),
                         query("miseg_iic_local_joint_ws_bytes", ub, k, h, w, pad, p)), dev)   # batched launch | per-head fallback
            px = sum((a1 - a0) * (b1 - b0) for a0, a1, b0, b1 in windows)
            nplanes = query("miseg_iic_local_planes_bytes", s, ub, k, h, w, pad) if _MI_PLANES and _mi_precision == MI_PRECISIONS["f16f8"] else 0
            if nplanes > 0:
                # the joint forward also writes every probability out as the backward's operand images (f16 hi, two e4m3 planes,
                # pixel-major): the backward's source rows are then plain memory -> LDS copies, and this node keeps the planes, not probs
                planes = torch.empty(nplanes, dtype=torch.uint8, device=dev)
                call("miseg_iic_local_joint_fwd_heads_planes", _stream(), _ptr(probs), s, ub, k, h, w, pad, _ptr(win), p, _ptr(raw), _ptr(ws),
                     ws.numel(), _ptr(planes), nplanes, work=(2.0 * k * k * t * t * ub * px * s, 2.0 * ub * k * px * (4 + 4.4) * s),
                     tag=f"iic_local_joint_fwd[p{pad}]")
            else:
                call("miseg_iic_local_joint_fwd_heads", _stream(), _ptr(probs), s, ub, k, h, w, pad, _ptr(win), p, _ptr(raw), _ptr(ws), ws.numel(),
                     _mi_precision, work=(2.0 * k * k * t * t * ub * px * s, 2.0 * ub * k * px * 4 * s), tag=f"iic_local_joint_fwd[p{pad}]")
        else:
            for i in range(s):
                _local_fwd(probs[i, :ub], probs[i, ub:], mask, pad, windows, win, raw[i])
        loss = scalar_out((s, p), dev)
        grad_raw = torch.empty_like(raw)
        _local_loss(raw, k, pad, s * p, lamda, loss, grad_raw)
        ctx.save_for_backward(planes if planes is not None else probs, mask, grad_raw, win)
        ctx.pad, ctx.windows, ctx.ub, ctx.planes, ctx.shape = pad, list(windows), ub, planes is not None, tuple(probs.shape)
        return loss

    @staticmethod
    def backward(ctx, gloss: Tensor):
        probs, mask, grad_raw, win = ctx.saved_tensors
        s, _, k, h, w = ctx.shape
        ub = ctx.ub
        gprob = torch.empty(ctx.shape, dtype=torch.float32, device=probs.device)
        if not _is_whole(ctx.windows, h, w):
            fill_zero(gprob)
        scale = gloss.contiguous().float()
        if mask is None:
            whole = _is_whole(ctx.windows, h, w)
            tt = (2 * ctx.pad + 1) ** 2
            bws = _ws(query("miseg_iic_local_bwd_ws_bytes", k, ctx.pad, len(ctx.windows) * s), probs.device)   # >= the per-head size
            for grp in colour_windows(ctx.windows):
                if len(grp) == len(ctx.windows):
                    gwin, ggrad, gscale = win, grad_raw, scale
                else:
                    dev = probs.device
                    idx = cached_const(("idx32", str(dev), tuple(grp)), lambda: torch.tensor(grp, dtype=torch.int32, device=dev))
                    gwin = cached_const(("gwin", str(dev), tuple(ctx.windows), tuple(grp)), lambda: win[idx.long()].contiguous())
                    npw, per = len(ctx.windows), grad_raw[0, 0].numel()
                    ggrad = torch.empty((s, len(grp)) + tuple(grad_raw.shape[2:]), dtype=torch.float32, device=dev)
                    gscale = torch.empty((s, len(grp)), dtype=torch.float32, device=dev)
                    call("miseg_gather_rows", _stream(), _ptr(grad_raw), _ptr(ggrad), s, npw, len(grp), _ptr(idx), per)
                    call("miseg_gather_rows", _stream(), _ptr(scale), _ptr(gscale), s, npw, len(grp), _ptr(idx), 1)
                px = sum((ctx.windows[i][1] - ctx.windows[i][0]) * (ctx.windows[i][3] - ctx.windows[i][2]) for i in grp)
                if ctx.planes:
                    call("miseg_iic_local_bwd_heads_planes", _stream(), _ptr(probs), probs.numel(), s, ub, k, h, w, ctx.pad, _ptr(gwin), len(grp),
                         _ptr(ggrad), _ptr(gscale), _ptr(gprob), 0 if whole else 1, _ptr(bws), bws.numel(),
                         work=(4.0 * k * k * tt * ub * px * s, 2.0 * ub * k * px * (4.4 + 4) * s), tag=f"iic_local_bwd[p{ctx.pad}]")
                else:
                    call("miseg_iic_local_bwd_heads", _stream(), _ptr(probs), s, ub, k, h, w, ctx.pad, _ptr(gwin), len(grp), _ptr(ggrad),
                         _ptr(gscale), _ptr(gprob), 0 if whole else 1, _mi_precision, _ptr(bws), bws.numel(),
                         work=(4.0 * k * k * tt * ub * px * s, 4.0 * ub * k * px * 4 * s), tag=f"iic_local_bwd[p{ctx.pad}]")
        else:
            for i in range(s):
                _local_bwd(probs[i, :ub], probs[i, ub:], mask, ctx.pad, ctx.windows, win, grad_raw[i], scale[i], gprob[i, :ub],
                           gprob[i, ub:])
        return gprob, None, None, None, None, None


def local_mi_heads(probs: Tensor, ub: int, pad: int, windows, lamda: float = 1.0, mask: Optional[Tensor] = None) -> Tensor:
    """loss[S][P] for probs[S, 2*UB, K, H, W] (see _LocalMIHeads)."""
    return _LocalMIHeads.apply(probs, int(ub), mask, int(pad), [tuple(int(v) for v in w) for w in windows], float(lamda))


def local_mi_losses(x: Tensor, y: Tensor, pad: int, windows, lamda: float = 1.0, mask: Optional[Tensor] = None) -> Tensor:
    return _LocalMI.apply(x, y, mask, int(pad), [tuple(int(v) for v in w) for w in windows], float(lamda))


def local_mi_raw_joint(x: Tensor, y: Tensor, pad: int, windows, mask: Optional[Tensor] = None, precision: Optional[str] = None) -> Tensor:
    """raw[P][T][T][K][K] only (no autograd) -- exposed for parity tests of the contraction."""
    _need_gpu(x, y, mask)
    x, y = x.contiguous().float(), y.contiguous().float()
    n, k, h, w = x.shape
    p = len(windows)
    win = torch.tensor(windows, dtype=torch.int32, device=x.device).view(p, 4)
    t = 2 * pad + 1
    raw = torch.empty(p, t, t, k, k, dtype=torch.float32, device=x.device)
    ws = _ws(query("miseg_iic_local_joint_ws_bytes", n, k, h, w, pad, p), x.device)
    call("miseg_iic_local_joint_fwd", _stream(), _ptr(x), _ptr(y), _ptr(mask), n, k, h, w, pad, _ptr(win), p, _ptr(raw),
         _ptr(ws), ws.numel(), _mi_precision if precision is None else MI_PRECISIONS[precision])
    return raw


# ------------------------------------------------------------------------------------------ global MI
class _GlobalMI(torch.autograd.Function):
    """IIDLoss for S sub-heads at once: x, y [S,N,K] -> (loss[S], loss_no_lamb[S], joint[S,K,K])."""

    @staticmethod
    def forward(ctx, x: Tensor, y: Tensor, lamb: float):
        _need_gpu(x, y)
        x, y = x.contiguous().float(), y.contiguous().float()
        s, n, k = x.shape
        loss = scalar_out((s,), x.device)
        loss_nl = torch.empty_like(loss)
        joint = torch.empty(s, k, k, dtype=torch.float32, device=x.device)
        call("miseg_iic_global_fwd", _stream(), _ptr(x), _ptr(y), s, n, k, float(lamb), _ptr(loss), _ptr(loss_nl), _ptr(joint))
        ctx.save_for_backward(x, y)
        ctx.lamb = float(lamb)
        ctx.mark_non_differentiable(loss_nl, joint)
        ctx.set_materialize_grads(False)      # (autograd would fill zero gradients for the two non-differentiable outputs: two launches)
        return loss, loss_nl, joint

    @staticmethod
    def backward(ctx, gloss, _g1, _g2):
        if gloss is None:
            return None, None, None
        x, y = ctx.saved_tensors
        s, n, k = x.shape
        gx, gy = torch.empty_like(x), torch.empty_like(y)
        up = gloss.contiguous().float()
        call("miseg_iic_global_bwd", _stream(), _ptr(x), _ptr(y), s, n, k, ctx.lamb, _ptr(up), _ptr(gx), _ptr(gy))
        return gx, gy, None


def global_mi(x: Tensor, y: Tensor, lamb: float = 1.0):
    return _GlobalMI.apply(x, y, float(lamb))


class _GradJoin:
    """Hand-over of an input gradient between two consumers of ONE feature map, so that autograd has nothing to add.

    The last decoder block's output feeds DeConv_1x1 (the logits) and the local-MI head.  Autograd would add their two 100 MB input
    gradients with an elementwise kernel on the step's critical path (right after the IIC chain; 45 us alone, up to 150 us beside
    the next tap's backward).  Instead the consumer whose backward runs first (``conv1x1``: it has its gradient as soon as the
    supervised / consistency losses are differentiated) ``offer``s the tensor it wrote; the head's backward ``take``s it, makes its
    stream wait for the writer, ACCUMULATES into it in its kernel epilogue (``miseg_head_local_bwd_acc``), makes the writer's
    stream wait for that kernel, and returns no gradient of its own.  No offer (another tap, another order, an unsupported shape):
    the head returns its gradient as before.  The protocol is symmetric -- whoever runs first returns its tensor to autograd AND
    offers it, whoever runs second and can accumulate takes it and returns None: at the 32-channel tap the head runs first and
    ``up_conv``'s pooled data gradient (``miseg_conv3x3_fwd_sumpool_acc``) is the one that adds.

    OFF unless ``MISEG_GRAD_JOIN=1``: once the per-step host synchronisation was gone and same-box repeats agreed to 0.02 ms, three
    alternations read 7.32 / 7.30 / 7.33 ms without, 7.34 / 7.34 / 7.36 with the join at the 16-channel tap only, 7.44 / 7.43 / 7.44 at the
    32-channel tap only, 7.41 with both -- the cross-stream waits the hand-over adds cost more than the add kernels it removes."""
    enabled = os.environ.get("MISEG_GRAD_JOIN", "0") == "1"
    only = os.environ.get("MISEG_GRAD_JOIN_ONLY", "")      # "16" / "32": join at taps of that width only (the measurements below)
    _offers: dict = {}

    @classmethod
    def clear(cls) -> None:
        cls._offers.clear()

    @classmethod
    def offer(cls, feature: Tensor, grad: Tensor) -> None:
        """Only inside a backward pass; the offers die with it (a storage address may be another tensor's in the next pass)."""
        if cls.enabled and feature.is_cuda and grad.shape == feature.shape and grad.dtype == feature.dtype:
            if not cls._offers:
                try:
                    torch.autograd.Variable._execution_engine.queue_callback(cls.clear)
                except RuntimeError:      # not called from a backward pass: no hand-over
                    return
            ev = torch.cuda.Event()
            ev.record()
            cls._offers[(feature.data_ptr(), tuple(feature.shape))] = (grad, ev, torch.cuda.current_stream(feature.device))

    @classmethod
    def take(cls, feature: Tensor):
        return cls._offers.pop((feature.data_ptr(), tuple(feature.shape)), None) if cls.enabled else None


class _GlobalMIPair(torch.autograd.Function):
    """IIDLoss for S sub-heads on prob[S, 2N, K] = [view 1 | view 2] along dim 1 (the layout the heads produce): the kernels read the
    two halves in place and write ONE gradient tensor -- no slice copies, no zero-filled halves added together by autograd."""

    @staticmethod
    def forward(ctx, prob: Tensor, lamb: float):
        _need_gpu(prob)
        prob = prob.contiguous().float()
        s, n2, k = prob.shape
        n = n2 // 2
        loss = scalar_out((s,), prob.device)
        loss_nl = torch.empty_like(loss)
        joint = torch.empty(s, k, k, dtype=torch.float32, device=prob.device)
        call("miseg_iic_global_fwd_pair", _stream(), _ptr(prob), s, n, k, float(lamb), _ptr(loss), _ptr(loss_nl), _ptr(joint))
        ctx.save_for_backward(prob)
        ctx.lamb = float(lamb)
        ctx.mark_non_differentiable(loss_nl, joint)
        ctx.set_materialize_grads(False)      # (autograd would fill zero gradients for the two non-differentiable outputs: two launches)
        return loss, loss_nl, joint

    @staticmethod
    def backward(ctx, gloss, _g1, _g2):
        if gloss is None:
            return None, None
        prob, = ctx.saved_tensors
        s, n2, k = prob.shape
        gprob = torch.empty_like(prob)
        call("miseg_iic_global_bwd_pair", _stream(), _ptr(prob), s, n2 // 2, k, ctx.lamb, _ptr(gloss.contiguous().float()), _ptr(gprob))
        return gprob, None


def global_mi_pair(prob: Tensor, lamb: float = 1.0):
    """(loss[S], loss_no_lamb[S], joint[S,K,K]) of IIDLoss(prob[:, :N], prob[:, N:]) for prob[S, 2N, K]."""
    if prob.shape[1] % 2:
        raise _cabi.MisegError(f"global_mi_pair: prob holds {prob.shape[1]} rows per sub-head, expected an even number (two views)")
    return _GlobalMIPair.apply(prob, float(lamb))


class _GlobalJoint(torch.autograd.Function):
    """compute_joint (ref iic_loss.py:74-94) for S pairs at once: x, y [S,N,K] -> joint [S,K,K]."""

    @staticmethod
    def forward(ctx, x: Tensor, y: Tensor, symmetric: bool):
        _need_gpu(x, y)
        x, y = x.contiguous().float(), y.contiguous().float()
        s, n, k = x.shape
        joint = torch.empty(s, k, k, dtype=torch.float32, device=x.device)
        call("miseg_iic_global_joint_fwd", _stream(), _ptr(x), _ptr(y), s, n, k, int(bool(symmetric)), _ptr(joint))
        ctx.save_for_backward(x, y, joint)
        ctx.symmetric = bool(symmetric)
        return joint

    @staticmethod
    def backward(ctx, gjoint):
        x, y, joint = ctx.saved_tensors
        s, n, k = x.shape
        gx, gy = torch.empty_like(x), torch.empty_like(y)
        gjoint = gjoint.contiguous().float()
        call("miseg_iic_global_joint_bwd", _stream(), _ptr(x), _ptr(y), s, n, k, int(ctx.symmetric), _ptr(joint), _ptr(gjoint), _ptr(gx), _ptr(gy))
        return gx, gy, None


def global_joint(x: Tensor, y: Tensor, symmetric: bool = True) -> Tensor:
    return _GlobalJoint.apply(x, y, bool(symmetric))


# ------------------------------------------------------------------------------------------ heads
def as_nhwc(t: Tensor) -> Tensor:
    """[N,C,H,W]-shaped tensor whose memory is NHWC (channels_last); converts if needed."""
    if t.dim() != 4:
        raise ValueError(f"expected NCHW-shaped tensor, got {tuple(t.shape)}")
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def empty_nhwc(n, c, h, w, dtype, device) -> Tensor:
    return torch.empty((n, c, h, w), dtype=dtype, device=device, memory_format=torch.channels_last)


def zeros_nhwc(n, c, h, w, dtype, device) -> Tensor:
    return torch.empty((n, c, h, w), dtype=dtype, device=device, memory_format=torch.channels_last).zero_()


def _stacked_grad(params, shape, device) -> Tensor:
    """Gradient buffer for a stacked [S, ...] head parameter: the flat-gradient slots themselves when the S parameters
    sit back to back there (gradslot.register_adjacent), else a fresh tensor."""
    from .gradslot import stacked_grad_slot
    slot = stacked_grad_slot(params) if params else None
    return slot.view(shape) if slot is not None else torch.empty(shape, dtype=torch.float32, device=device)


_HEAD_RECOMPUTE = os.environ.get("MISEG_HEAD_RECOMPUTE", "1") != "0"   # 0: the head backward reads the saved probabilities


class _LocalHead(torch.autograd.Function):
    """S x (1x1 conv + channel softmax) with fused sample gather + flip replay -> prob [S,M,K,H,W]."""

    @staticmethod
    def forward(ctx, feat: Tensor, w: Tensor, b: Tensor, src: Tensor, flips: Optional[Tensor], temperature: float):
        _need_gpu(feat, w, b, src, flips)
        feat = as_nhwc(feat)
        bsz, c, h, wd = feat.shape
        s, k, _ = w.shape
        m = src.numel()
        ctx.stack_params = (getattr(w, "_miseg_stack_params", None), getattr(b, "_miseg_stack_params", None))
        w, b = w.contiguous().float(), b.contiguous().float()
        prob = torch.empty(s, m, k, h, wd, dtype=torch.float32, device=feat.device)
        viol = zero_counter(feat.device) if k <= 32 else None
        _LocalHead.last_violations = viol
        ctx.sink = getattr(feat, "_miseg_grad_sink", None)
        call("miseg_head_local_fwd", _stream(), _DT[feat.dtype], _ptr(feat), bsz, h, wd, c, _ptr(src), _ptr(flips), m, _ptr(w), _ptr(b),
             s, k, float(temperature), _ptr(prob), 2e-4, _ptr(viol), work=(2.0 * s * k * c * m * h * wd, (s * k * 4.0 + c * feat.element_size()) * m * h * wd),
             tag=f"head_local_fwd[c{c}]")
        # the shipped top tap: the backward computes the probabilities again from the features (bit-equal), so the node does not hold on
        # to them -- with the local-MI node keeping operand planes instead (see _LocalMIHeads) the fp32 tensor dies after the joint forward
        ctx.recompute = bool(_HEAD_RECOMPUTE and not _GradJoin.enabled and
                             query("miseg_head_local_bwd_recompute_supported", _DT[feat.dtype], c, s, k))
        ctx.save_for_backward(feat, w, src, flips, feat.new_empty(0) if ctx.recompute else prob, b)
        ctx.temperature = float(temperature)
        ctx.src_range = getattr(src, "_miseg_range", None)
        return prob

    @staticmethod
    def backward(ctx, gprob: Tensor):
        feat, w, src, flips, prob, b = ctx.saved_tensors
        bsz, c, h, wd = feat.shape
        s, k, _ = w.shape
        m = src.numel()
        gprob = gprob.contiguous().float()
        gfeat = None
        joined = None
        if ctx.needs_input_grad[0] and query("miseg_head_local_bwd_acc_supported", _DT[feat.dtype], c, s, k) and \
                (not _GradJoin.only or _GradJoin.only == str(c)):
            joined = _GradJoin.take(feat)
        if joined is not None:
            # the other consumer of this feature already wrote its input gradient: add ours to it in the kernel epilogue
            gsum, written, writer = joined
            cur = torch.cuda.current_stream(feat.device)
            cur.wait_event(written)
            gsum.record_stream(cur)
            gw = _stacked_grad(ctx.stack_params[0], w.shape, feat.device)
            gb = _stacked_grad(ctx.stack_params[1], (s, k), feat.device)
            ws = _ws(query("miseg_head_local_bwd_ws_bytes", m, h, wd, c, s, k), feat.device)
            call("miseg_head_local_bwd_acc", _stream(), _DT[feat.dtype], _ptr(feat), bsz, h, wd, c, _ptr(src), _ptr(flips), m, _ptr(w), s, k,
                 ctx.temperature, _ptr(prob), _ptr(gprob), _ptr(gsum), _ptr(gw), _ptr(gb), _ptr(ws), ws.numel(),
                 work=(4.0 * s * k * c * m * h * wd, (3 * s * k * 4.0 + 3 * c * feat.element_size()) * m * h * wd), tag=f"head_local_bwd[c{c}]")
            if writer != cur:
                done = torch.cuda.Event()
                done.record(cur)
                writer.wait_event(done)       # whoever reads the joined gradient next does so on the writer's stream
            return None, gw, gb, None, None, None
        sink = ctx.sink if not _GradJoin.enabled else None
        compact, n0, n1 = False, 0, bsz
        if ctx.needs_input_grad[0]:
            rng = ctx.src_range
            if rng is not None and feat.dtype in (torch.bfloat16, torch.float16) and k == 20 and s == 5 and c in (16, 32) and 0 <= rng[0] <= rng[1] <= bsz:
                # the shipped kernels store (not accumulate) every pixel and channel of the rows in src = [start, stop) and touch no other
                if sink is not None and rng[0] < rng[1]:
                    # ... so a COMPACT gradient of those rows is all there is: the producing layer's BatchNorm backward adds it in its
                    # loaders (unet_ops._GradSink) -- no zero fill of the other rows, no add kernel over the whole batch
                    n0, n1 = rng
                    gfeat = torch.empty((n1 - n0, c, h, wd), dtype=feat.dtype, device=feat.device, memory_format=torch.channels_last)
                    compact = True
                else:
                    gfeat = torch.empty((bsz, c, h, wd), dtype=feat.dtype, device=feat.device, memory_format=torch.channels_last)
                    if rng[0] > 0:
                        fill_zero(gfeat[:rng[0]])
                    if rng[1] < bsz:
                        fill_zero(gfeat[rng[1]:])
            else:
                gfeat = fill_zero(empty_nhwc(bsz, c, h, wd, feat.dtype, feat.device))
        gw = _stacked_grad(ctx.stack_params[0], w.shape, feat.device)
        gb = _stacked_grad(ctx.stack_params[1], (s, k), feat.device)
        ws = _ws(query("miseg_head_local_bwd_ws_bytes", m, h, wd, c, s, k), feat.device)
        work = (4.0 * s * k * c * m * h * wd, (3 * s * k * 4.0 + 2 * c * feat.element_size()) * m * h * wd)
        if ctx.recompute:
            # the kernel computes the probabilities again from the features (bit-equal to the forward's): it reads gprob only
            call("miseg_head_local_bwd_recompute", _stream(), _DT[feat.dtype], _ptr(feat), bsz, h, wd, c, _ptr(src), _ptr(flips), m, _ptr(w),
                 _ptr(b), s, k, ctx.temperature, _ptr(gprob), _ptr(gfeat), n0, _ptr(gw), _ptr(gb), _ptr(ws), ws.numel(),
                 work=(work[0] + 2.0 * s * k * c * m * h * wd, work[1] - s * k * 4.0 * m * h * wd), tag=f"head_local_bwd[c{c}]")
        elif compact:
            call("miseg_head_local_bwd_rows", _stream(), _DT[feat.dtype], _ptr(feat), bsz, h, wd, c, _ptr(src), _ptr(flips), m, _ptr(w), s, k,
                 ctx.temperature, _ptr(prob), _ptr(gprob), _ptr(gfeat), n0, _ptr(gw), _ptr(gb), _ptr(ws), ws.numel(), work=work,
                 tag=f"head_local_bwd[c{c}]")
        else:
            call("miseg_head_local_bwd", _stream(), _DT[feat.dtype], _ptr(feat), bsz, h, wd, c, _ptr(src), _ptr(flips), m, _ptr(w), s, k,
                 ctx.temperature, _ptr(prob), _ptr(gprob), _ptr(gfeat), _ptr(gw), _ptr(gb), _ptr(ws), ws.numel(), work=work,
                 tag=f"head_local_bwd[c{c}]")
        if gfeat is not None and sink is not None:
            sink.put(gfeat, n0, n1)          # handed to the producing layer's backward; autograd gets no gradient from this edge
            return None, gw, gb, None, None, None
        if gfeat is not None:
            _GradJoin.offer(feat, gfeat)     # a later consumer of the same feature (e.g. up_conv's pooled data gradient) may add into it
        return gfeat, gw, gb, None, None, None


def local_head(feat: Tensor, w: Tensor, b: Tensor, src: Tensor, flips: Optional[Tensor], temperature: float = 1.0) -> Tensor:
    """prob[S, M, K, H, W].  The kernel also counts the positions whose K probabilities do not sum to one; the count rides
    on the result (``_miseg_simplex`` = (axis, device counter, tensor version)) so that a later ``checks.assert_simplex`` of
    this very tensor costs nothing."""
    _LocalHead.last_violations = None
    prob = _LocalHead.apply(feat, w, b, src, flips, float(temperature))
    if _LocalHead.last_violations is not None:
        prob._miseg_simplex = (2, _LocalHead.last_violations, prob._version)
        _LocalHead.last_violations = None
    return prob


class _GlobalHead(torch.autograd.Function):
    """S x (global avg pool -> Linear -> softmax) -> prob [S,M,K]."""

    @staticmethod
    def forward(ctx, feat: Tensor, w: Tensor, b: Tensor, src: Tensor, temperature: float):
        _need_gpu(feat, w, b, src)
        feat = as_nhwc(feat)
        bsz, c, h, wd = feat.shape
        s, k, _ = w.shape
        m = src.numel()
        ctx.stack_params = (getattr(w, "_miseg_stack_params", None), getattr(b, "_miseg_stack_params", None))
        w, b = w.contiguous().float(), b.contiguous().float()
        pooled = torch.empty(m, c, dtype=torch.float32, device=feat.device)
        prob = torch.empty(s, m, k, dtype=torch.float32, device=feat.device)
        call("miseg_head_global_fwd", _stream(), _DT[feat.dtype], _ptr(feat), bsz, h, wd, c, _ptr(src), m, _ptr(w), _ptr(b), s, k,
             float(temperature), _ptr(pooled), _ptr(prob))
        ctx.save_for_backward(w, src, pooled, prob)
        ctx.meta = (bsz, c, h, wd, feat.dtype, float(temperature))
        ctx.sink = getattr(feat, "_miseg_grad_sink", None)
        ctx.src_range = getattr(src, "_miseg_range", None)
        return prob

    @staticmethod
    def backward(ctx, gprob: Tensor):
        w, src, pooled, prob = ctx.saved_tensors
        bsz, c, h, wd, dtype, temperature = ctx.meta
        s, k, _ = w.shape
        m = src.numel()
        gprob = gprob.contiguous().float()
        gfeat, compact, n0, n1 = None, False, 0, bsz
        sink, rng = ctx.sink, ctx.src_range
        if ctx.needs_input_grad[0]:
            if sink is not None and rng is not None and 0 <= rng[0] < rng[1] <= bsz and rng[1] - rng[0] == m:
                # the kernel writes every element of the rows in src = [start, stop) and nothing else: a compact gradient (_LocalHead)
                n0, n1 = rng
                gfeat = empty_nhwc(n1 - n0, c, h, wd, dtype, w.device)
                compact = True
            else:
                gfeat = fill_zero(empty_nhwc(bsz, c, h, wd, dtype, w.device))
        gw = _stacked_grad(ctx.stack_params[0], w.shape, w.device)
        gb = _stacked_grad(ctx.stack_params[1], (s, k), w.device)
        dz = torch.empty(s * m * k, dtype=torch.float32, device=w.device)
        if compact:
            call("miseg_head_global_bwd_rows", _stream(), _DT[dtype], bsz, h, wd, c, _ptr(src), m, _ptr(w), s, k, temperature, _ptr(pooled),
                 _ptr(prob), _ptr(gprob), _ptr(gfeat), n0, _ptr(gw), _ptr(gb), _ptr(dz))
        else:
            call("miseg_head_global_bwd", _stream(), _DT[dtype], bsz, h, wd, c, _ptr(src), m, _ptr(w), s, k, temperature, _ptr(pooled),
                 _ptr(prob), _ptr(gprob), _ptr(gfeat), _ptr(gw), _ptr(gb), _ptr(dz))
        if gfeat is not None and sink is not None:
            sink.put(gfeat, n0, n1)
            return None, gw, gb, None, None
        return gfeat, gw, gb, None, None


def global_head(feat: Tensor, w: Tensor, b: Tensor, src: Tensor, temperature: float = 1.0) -> Tensor:
    return _GlobalHead.apply(feat, w, b, src, float(temperature))


# ------------------------------------------------------------------------------------------ head variants (mlp / normalize)
class _LocalHeadVar(torch.autograd.Function):
    """LocalClusterHead with head_type='mlp' and/or normalize=True (csrc/heads_var.hip) -> prob [S,M,K,H,W].
    hid == 0: single layer (w1 [S,K,C], b1 [S,K]; w2, b2 are dummies).  The backward recomputes the forward from the features."""

    @staticmethod
    def forward(ctx, feat, w1, b1, w2, b2, src, flips, temperature: float, normalize: bool, hid: int):
        _need_gpu(feat, w1, b1, src, flips)
        feat = as_nhwc(feat)
        bsz, c, h, wd = feat.shape
        s = w1.shape[0]
        k = (w2 if hid else w1).shape[1]
        m = src.numel()
        ctx.stacks = tuple(getattr(t, "_miseg_stack_params", None) for t in (w1, b1, w2, b2))
        w1, b1 = w1.contiguous().float(), b1.contiguous().float()
        w2, b2 = (w2.contiguous().float(), b2.contiguous().float()) if hid else (None, None)
        prob = torch.empty(s, m, k, h, wd, dtype=torch.float32, device=feat.device)
        call("miseg_head_local_var_fwd", _stream(), _DT[feat.dtype], _ptr(feat), bsz, h, wd, c, _ptr(src), _ptr(flips), m, _ptr(w1), _ptr(b1),
             hid, _ptr(w2), _ptr(b2), s, k, float(temperature), int(bool(normalize)), _ptr(prob))
        ctx.save_for_backward(feat, w1, b1, w2, b2, src, flips)
        ctx.cfg = (float(temperature), bool(normalize), int(hid), k)
        return prob

    @staticmethod
    def backward(ctx, gprob):
        feat, w1, b1, w2, b2, src, flips = ctx.saved_tensors
        temperature, normalize, hid, k = ctx.cfg
        bsz, c, h, wd = feat.shape
        s, m, dev = w1.shape[0], src.numel(), feat.device
        gprob = gprob.contiguous().float()
        gfeat = zeros_nhwc(bsz, c, h, wd, feat.dtype, dev) if ctx.needs_input_grad[0] else None
        gw1 = _stacked_grad(ctx.stacks[0], w1.shape, dev)
        gb1 = _stacked_grad(ctx.stacks[1], b1.shape, dev)
        gw2 = _stacked_grad(ctx.stacks[2], w2.shape, dev) if hid else None
        gb2 = _stacked_grad(ctx.stacks[3], b2.shape, dev) if hid else None
        ws = _ws(query("miseg_head_local_var_bwd_ws_bytes", m, h, wd, c, hid, s, k), dev)
        call("miseg_head_local_var_bwd", _stream(), _DT[feat.dtype], _ptr(feat), bsz, h, wd, c, _ptr(src), _ptr(flips), m, _ptr(w1), _ptr(b1),
             hid, _ptr(w2), _ptr(b2), s, k, temperature, int(normalize), _ptr(gprob), _ptr(gfeat), _ptr(gw1), _ptr(gb1), _ptr(gw2), _ptr(gb2),
             _ptr(ws), ws.numel())
        return gfeat, gw1, gb1, gw2, gb2, None, None, None, None, None


def local_head_var(feat, w1, b1, w2, b2, src, flips, temperature=1.0, normalize=False) -> Tensor:
    """w2 is None for the single-layer head.  w1 [S,R,C] (R = hidden width or K), w2 [S,K,HID]."""
    hid = 0 if w2 is None else int(w1.shape[1])
    if w2 is None:
        w2 = b2 = torch.empty(0, device=feat.device)
    return _LocalHeadVar.apply(feat, w1, b1, w2, b2, src, flips, float(temperature), bool(normalize), hid)


class _GlobalHeadVar(torch.autograd.Function):
    """ClusterHead with head_type='mlp' and/or normalize=True -> prob [S,M,K]."""

    @staticmethod
    def forward(ctx, feat, w1, b1, w2, b2, src, temperature: float, normalize: bool, hid: int):
        _need_gpu(feat, w1, b1, src)
        feat = as_nhwc(feat)
        bsz, c, h, wd = feat.shape
        s = w1.shape[0]
        k = (w2 if hid else w1).shape[1]
        m = src.numel()
        ctx.stacks = tuple(getattr(t, "_miseg_stack_params", None) for t in (w1, b1, w2, b2))
        w1, b1 = w1.contiguous().float(), b1.contiguous().float()
        w2, b2 = (w2.contiguous().float(), b2.contiguous().float()) if hid else (None, None)
        pooled = torch.empty(m, c, dtype=torch.float32, device=feat.device)
        prob = torch.empty(s, m, k, dtype=torch.float32, device=feat.device)
        call("miseg_head_global_var_fwd", _stream(), _DT[feat.dtype], _ptr(feat), bsz, h, wd, c, _ptr(src), m, _ptr(w1), _ptr(b1), hid, _ptr(w2),
             _ptr(b2), s, k, float(temperature), int(bool(normalize)), _ptr(pooled), _ptr(prob))
        ctx.save_for_backward(w1, b1, w2, b2, src, pooled)
        ctx.cfg = (float(temperature), bool(normalize), int(hid), k, bsz, c, h, wd, feat.dtype)
        return prob

    @staticmethod
    def backward(ctx, gprob):
        w1, b1, w2, b2, src, pooled = ctx.saved_tensors
        temperature, normalize, hid, k, bsz, c, h, wd, dtype = ctx.cfg
        s, m, dev = w1.shape[0], src.numel(), w1.device
        gprob = gprob.contiguous().float()
        gfeat = zeros_nhwc(bsz, c, h, wd, dtype, dev) if ctx.needs_input_grad[0] else None
        gw1 = _stacked_grad(ctx.stacks[0], w1.shape, dev)
        gb1 = _stacked_grad(ctx.stacks[1], b1.shape, dev)
        gw2 = _stacked_grad(ctx.stacks[2], w2.shape, dev) if hid else None
        gb2 = _stacked_grad(ctx.stacks[3], b2.shape, dev) if hid else None
        dpool = torch.empty(s * m * c, dtype=torch.float32, device=dev)
        call("miseg_head_global_var_bwd", _stream(), _DT[dtype], bsz, h, wd, c, _ptr(src), m, _ptr(w1), _ptr(b1), hid, _ptr(w2), _ptr(b2), s, k,
             temperature, int(normalize), _ptr(pooled), _ptr(gprob), _ptr(gfeat), _ptr(gw1), _ptr(gb1), _ptr(gw2), _ptr(gb2), _ptr(dpool))
        return gfeat, gw1, gb1, gw2, gb2, None, None, None, None


def global_head_var(feat, w1, b1, w2, b2, src, temperature=1.0, normalize=False) -> Tensor:
    hid = 0 if w2 is None else int(w1.shape[1])
    if w2 is None:
        w2 = b2 = torch.empty(0, device=feat.device)
    return _GlobalHeadVar.apply(feat, w1, b1, w2, b2, src, float(temperature), bool(normalize), hid)


# ------------------------------------------------------------------------------------------ pixel losses
def _logits_nhwc(t: Tensor) -> Tensor:
    t = as_nhwc(t)
    return t if t.dtype == torch.float32 else t.float()


class _SoftmaxKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, labels: Tensor):
        _need_gpu(logits, labels)
        raw_logits = logits
        logits = _logits_nhwc(logits)
        n, c, h, w = logits.shape
        labels = labels.contiguous().view(n, h, w)
        if labels.dtype != torch.int64:
            labels = labels.long()
        dev = logits.device
        loss = scalar_out((), dev)
        glogits = empty_nhwc(n, c, h, w, torch.float32, dev)
        bad = zero_counter(dev, (1,))
        ws = _ws(query("miseg_loss_ws_bytes", n, h, w), dev)
        call("miseg_softmax_kl", _stream(), _ptr(logits), _ptr(labels), n, h, w, c, None, _ptr(loss), _ptr(glogits), _ptr(bad),
             _ptr(ws), ws.numel())
        ctx.save_for_backward(glogits)
        ctx.bad = bad
        ctx.part = _split_part_of(raw_logits, logits)
        return loss

    @staticmethod
    def backward(ctx, g: Tensor):
        (glogits,) = ctx.saved_tensors
        return _scaled_grad(ctx.part, glogits, g), None


def softmax_kl(logits: Tensor, labels: Tensor, check_labels: bool = True) -> Tensor:
    """KL_div(softmax(logits), class2one_hot(labels)) with reduction='mean'."""
    return _SoftmaxKL.apply(logits, labels)


class _SoftmaxMSE(torch.autograd.Function):
    entry = "miseg_softmax_mse"

    @classmethod
    def forward(cls, ctx, a: Tensor, b: Tensor, flips: Optional[Tensor]):
        _need_gpu(a, b, flips)
        raw_a = a
        a, b = _logits_nhwc(a), _logits_nhwc(b.detach())
        n, c, h, w = a.shape
        dev = a.device
        loss = scalar_out((), dev)
        ga = empty_nhwc(n, c, h, w, torch.float32, dev)
        ws = _ws(query("miseg_loss_ws_bytes", n, h, w), dev)
        call(cls.entry, _stream(), _ptr(a), _ptr(b), _ptr(flips), n, h, w, c, None, _ptr(loss), _ptr(ga), _ptr(ws), ws.numel())
        ctx.save_for_backward(ga)
        ctx.part = _split_part_of(raw_a, a)
        return loss

    @staticmethod
    def backward(ctx, g: Tensor):
        (ga,) = ctx.saved_tensors
        return _scaled_grad(ctx.part, ga, g), None, None


def softmax_mse(a: Tensor, b: Tensor, flips: Optional[Tensor] = None) -> Tensor:
    """mean((softmax(a) - softmax(flip(b)).detach())**2); ``flips`` replays the per-sample flip on b."""
    return _SoftmaxMSE.apply(a, b, flips)


class _SoftmaxKLCons(_SoftmaxMSE):
    entry = "miseg_softmax_klcons"


def softmax_kl_consistency(a: Tensor, b: Tensor, flips: Optional[Tensor] = None) -> Tensor:
    """KL_div()(softmax(a), softmax(flip(b)).detach()) -- the `UDARegCriterion.name: kl` consistency term, fused like softmax_mse."""
    return _SoftmaxKLCons.apply(a, b, flips)


def _flip_raw(x: Tensor, flips: Tensor) -> Tensor:
    _need_gpu(x, flips)
    if x.element_size() not in (2, 4, 8):
        raise ValueError("flip: element size must be 2, 4 or 8 bytes")
    out = torch.empty_like(x)
    n, c, h, w = x.shape
    ist = torch.tensor(x.stride(), dtype=torch.int64)   # host arrays, read on the host side of the C call
    ost = torch.tensor(out.stride(), dtype=torch.int64)
    call("miseg_flip", _stream(), _ptr(x), _ptr(out), n, c, h, w, ist.data_ptr(), ost.data_ptr(), x.element_size(), _ptr(flips))
    return out


class _Flip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, flips: Tensor):
        ctx.save_for_backward(flips)
        return _flip_raw(x, flips)

    @staticmethod
    def backward(ctx, g: Tensor):
        (flips,) = ctx.saved_tensors
        return _flip_raw(g, flips), None  # a flip is its own inverse


def flip(x: Tensor, flips: Tensor) -> Tensor:
    """Per-sample H/W flip of an [N,C,H,W]-shaped tensor (any strides; 2/4/8-byte elements), bit-exact."""
    return _Flip.apply(x, flips) if x.requires_grad else _flip_raw(x, flips)


def argmax_dice(logits: Tensor, labels: Optional[Tensor], want_pred: bool = True, to_host: bool = False):
    """argmax over channels and per-sample per-class (intersection, union) int64 counts.  ``to_host``: inside a training iteration,
    write the counts into the iteration's read-back block (``stepio``) so that they travel with its scalars."""
    _need_gpu(logits, labels)
    logits = _logits_nhwc(logits)
    n, c, h, w = logits.shape
    dev = logits.device
    pred = torch.empty(n, h, w, dtype=torch.int64, device=dev) if want_pred else None
    inter = uni = None
    if labels is not None:
        labels = labels.contiguous().view(n, h, w).long()
        io = stepio.CURRENT if to_host else None
        if io is not None and io.device == dev:          # the iteration's one device -> host copy carries them
            inter, uni = io.out("inter", (n, c), torch.int64), io.out("union", (n, c), torch.int64)
        else:
            inter = torch.empty(n, c, dtype=torch.int64, device=dev)
            uni = torch.empty(n, c, dtype=torch.int64, device=dev)
    call("miseg_argmax_dice", _stream(), _ptr(logits), _ptr(labels), n, h, w, c, _ptr(pred), _ptr(inter), _ptr(uni))
    return pred, inter, uni


# ------------------------------------------------------------------------------------------ split with a layout-preserving backward
_SPLIT_MEMCPY = False   # True: assemble slice gradients with hipMemcpy (slower on a busy device)


class _SplitHolder:
    """Shared by the parts of one ``split_rows``: the batch gradient buffer that the split's backward returns.  A loss kernel whose
    input IS a part writes its scaled gradient straight into that part's rows (``_scaled_grad``: one launch, where autograd would
    multiply into a temporary and the split's backward would copy it over); the backward then only zero-fills the parts without a loss."""
    __slots__ = ("meta", "sizes", "out")

    def __init__(self, x: Tensor, sizes):
        cl = x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
        self.meta, self.sizes, self.out = (tuple(x.shape), x.dtype, x.device, cl), list(sizes), None

    def buffer(self) -> Tensor:
        if self.out is None:
            shape, dtype, device, cl = self.meta
            self.out = torch.empty(shape, dtype=dtype, device=device, memory_format=torch.channels_last if cl else torch.contiguous_format)
        return self.out

    def rows(self, idx: int) -> Tensor:
        return self.buffer().narrow(0, sum(self.sizes[:idx]), self.sizes[idx])


def _split_part_of(given: Tensor, used: Tensor):
    """(holder, index) if ``used`` -- the tensor the kernel read -- is, unconverted, a part of a split_rows."""
    tag = getattr(given, "_miseg_split", None)
    if tag is None or used.data_ptr() != given.data_ptr() or used.dtype != torch.float32:
        return None
    return tag


def _scaled_grad(part, grad: Tensor, g: Tensor) -> Tensor:
    """``grad * g`` for a 0-d upstream gradient ``g``; for a part of a split_rows, written into the part's rows of the batch
    gradient by one library launch (scale read from device memory) and returned as that view."""
    if part is not None and g.dim() == 0 and g.dtype == torch.float32 and g.is_cuda and grad.dtype == torch.float32 and grad.numel() % 4 == 0:
        holder, idx = part
        dst = holder.rows(idx)
        if dst.dtype == grad.dtype and dst.shape == grad.shape and dst.stride() == grad.stride():
            call("miseg_assemble_rows", _stream(), _ptr(dst), _ptr(grad), _ptr(g), grad.numel(), None, None, 0, None, None, 0)
            return dst
    return grad * g


class _SplitRows(torch.autograd.Function):
    """``torch.split(x, sizes, dim=0)`` whose backward assembles the gradient directly in x's memory format.

    The logits leave the network as one NHWC batch [labeled | unlabeled | flipped unlabeled] and are split for the losses
    (ref semi_seg/epocher.py:150-152).  Autograd's own split backward concatenates the per-part gradients, and a part without a
    loss (the detached UDA branch) arrives as NCHW zeros: the concatenation then comes out NCHW and the next kernel's NHWC view
    of it is a 4-channel transposing copy -- measured 1.35 ms per step for 50 MB.  Here: one NHWC buffer; parts whose loss kernel
    wrote its rows already (``_scaled_grad``) cost nothing, absent parts are zero-filled, anything else is copied in."""

    @staticmethod
    def forward(ctx, x: Tensor, holder, *sizes: int):
        ctx.sizes = sizes
        ctx.set_materialize_grads(False)    # a part without a loss stays None (materialised it would be NCHW zeros: a transposing copy)
        ctx.meta = (x.shape, x.dtype, x.device, x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))
        outs, o = [], 0
        for n in sizes:
            outs.append(x.narrow(0, o, n))
            o += n
        ctx.holder = holder
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        shape, dtype, device, cl = ctx.meta
        holder = ctx.holder
        if holder is not None:
            out, holder.out = holder.buffer(), None
        else:
            out = torch.empty(shape, dtype=dtype, device=device, memory_format=torch.channels_last if cl else torch.contiguous_format)
        o = 0
        for n, g in zip(ctx.sizes, grads):
            part = out.narrow(0, o, n)
            if g is None:
                if part.is_cuda and n:
                    fill_zero(part)
                else:
                    part.zero_()
            elif g.data_ptr() == part.data_ptr() and g.shape == part.shape and g.stride() == part.stride():
                pass                              # its loss kernel's backward wrote these rows (_scaled_grad)
            elif _SPLIT_MEMCPY:
                part.copy_(g)
            else:
                torch.mul(g, 1.0, out=part)   # an elementwise kernel: copy_ of two dense tensors is a hipMemcpyDtoD, seen at 1.5 ms for 16 MB
            o += n
        return (out,) + (None,) * (len(ctx.sizes) + 1)


def split_rows(x: Tensor, sizes) -> tuple:
    holder = _SplitHolder(x, [int(n) for n in sizes]) if x.is_cuda and x.requires_grad else None
    outs = _SplitRows.apply(x, holder, *[int(n) for n in sizes])
    if holder is not None:
        for i, t in enumerate(outs):
            t._miseg_split = (holder, i)
    return outs


def cat_flip(a: Tensor, b: Tensor, flips: Tensor, stem_dtype=None) -> Tensor:
    """``torch.cat([a, b, flip(b)])`` along dim 0 in one launch (ref semi_seg/epocher.py:148-153: the network's input batch).
    ``stem_dtype`` (bfloat16 / float16, one-channel fp32 images): the same launch also writes the first convolution's operand -- the
    batch cast to that type and padded to one channel vector -- and hangs it on the result (``_miseg_stem``) for ``unet_ops.stem_input``."""
    _need_gpu(a, b, flips)
    a, b = a.contiguous(), b.contiguous()
    assert a.dtype == b.dtype and a.shape[1:] == b.shape[1:] and a.element_size() == 4 and a.dim() == 4, (a.shape, b.shape, a.dtype)
    na, nb = a.shape[0], b.shape[0]
    out = torch.empty((na + 2 * nb,) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
    from . import unet_ops as _uo
    if stem_dtype in (torch.bfloat16, torch.float16) and a.dtype == torch.float32 and a.shape[1] == 1 and not _uo._STEM_KERNELS:
        pad = empty_nhwc(na + 2 * nb, 8, a.shape[2], a.shape[3], stem_dtype, a.device)
        call("miseg_cat_flip_pad", _stream(), _ptr(a), na, _ptr(b), nb, a.shape[2], a.shape[3], _ptr(flips), _ptr(out), _DT[stem_dtype], _ptr(pad))
        out._miseg_stem = (stem_dtype, pad)
        return out
    call("miseg_cat_flip", _stream(), _ptr(a), na, _ptr(b), nb, a.shape[1], a.shape[2], a.shape[3], _ptr(flips), _ptr(out))
    return out
