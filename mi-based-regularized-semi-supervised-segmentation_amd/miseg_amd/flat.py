"""Flat parameter / gradient storage and the fused Adam that runs on it.

All trainable tensors of the step (U-Net + projector heads, 2.19 M fp32 values at the shipped config)
are re-pointed into ONE contiguous fp32 buffer, their ``.grad`` into a second one.  That gives
  * the optimiser a single HIP launch per step (``miseg_adam_step``; ref semi_seg/trainer.py:67-72,179-184
    configures torch.optim.Adam with L2 weight decay -- same update rule, same state_dict layout);
  * data-parallel training a handful of large contiguous RCCL all-reduces instead of ~70 small ones
    (``miseg_amd.ddp.GradReducer`` buckets are slices of the flat gradient).
``torch.optim.Optimizer`` is subclassed so schedulers, ``param_groups`` and checkpoints behave as usual.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional

import torch
from torch import Tensor

from . import unet_ops
from .gradslot import grad_slot  # noqa: F401  (re-exported)


class FlatBuffers:
    """Owns the flat param / grad buffers of a list of parameters (all on one GPU, fp32).

    Gradient protocol: ``zero_grad()`` drops every ``.grad`` (no fill kernel) and opens the slots; during backward our
    conv / BN nodes write straight into their slots (``grad_slot``), every other gradient arrives wherever autograd put it;
    ``collect()`` -- called by the optimiser step and by the data-parallel reducer before a bucket is reduced -- moves
    stragglers into the flat buffer and zero-fills the slots of parameters that received no gradient."""

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.flat_param: Optional[Tensor] = None
        self.flat_grad: Optional[Tensor] = None
        self.offsets: List[int] = []
        self.slots: List[Tensor] = []
        self.total = 0

    def valid(self) -> bool:
        if self.flat_param is None:
            return False
        first, last = self.params[0], self.params[-1]
        base = self.flat_param.data_ptr()
        return first.data_ptr() == base and last.data_ptr() == base + 4 * self.offsets[-1]

    def build(self) -> None:
        dev = self.params[0].device  # any device: the reducer's bucketing is also exercised on CPU/gloo in the tests;
        # the fused Adam kernel itself is GPU-only and raises on CPU tensors
        self.params = self._layout_order(self.params)   # adjacency groups (cluster-head sub-heads) back to back
        self.offsets, off = [], 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise RuntimeError("flat buffers need fp32 parameters on a single device")
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned
        self.total = off
        flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.slots = []
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = flat_p[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                gview = flat_g[o:o + p.numel()].view_as(p)
                if p.grad is not None:
                    gview.copy_(p.grad)
                    p.grad = gview
                p.data = view
                p._miseg_grad_slot = gview
                p._miseg_grad_claimed = True   # slots open at zero_grad()
                self.slots.append(gview)
        self._offset_by_id = {id(p): o for p, o in zip(self.params, self.offsets)}
        self.flat_param, self.flat_grad = flat_p, flat_g
        unet_ops.PACK_CACHE.invalidate()

    @staticmethod
    def _layout_order(params):
        from .gradslot import adjacent_groups
        mine = {id(p) for p in params}
        group_of = {}
        for grp in adjacent_groups():
            if all(id(p) in mine for p in grp) and all(p.numel() % 4 == 0 for p in grp):
                for p in grp:
                    group_of.setdefault(id(p), grp)
        out, seen = [], set()
        for p in params:
            if id(p) in seen:
                continue
            for q in group_of.get(id(p), [p]):
                if id(q) not in seen:
                    seen.add(id(q))
                    out.append(q)
        return out

    def offset_of(self, p) -> int:
        return self._offset_by_id[id(p)]

    def ensure(self) -> None:
        if not self.valid():
            self.build()

    def zero_grad(self) -> None:
        self.ensure()
        for p in self.params:
            p.grad = None
            p._miseg_grad_claimed = False

    def collect(self, lo: int = 0, hi: Optional[int] = None) -> None:
        """Make ``flat_grad[offsets[lo]:offsets[hi])`` hold the gradients of params[lo:hi] (see class docstring)."""
        self.ensure()
        hi = len(self.params) if hi is None else hi
        if self.flat_grad.is_cuda:
            from .unet_ops import join_wgrad_streams
            join_wgrad_streams()          # weight gradients are written into the slots from a side stream
        with torch.no_grad():
            for i in range(lo, hi):
                p, slot = self.params[i], self.slots[i]
                g = p.grad
                if g is None:
                    if slot.is_cuda:
                        from .ops import fill_zero
                        fill_zero(slot)                # library launches: they are on the launch tape
                    else:
                        slot.zero_()
                elif g.data_ptr() != slot.data_ptr():
                    if slot.is_cuda and g.dtype == slot.dtype and g.shape == slot.shape and g.is_contiguous() and g.data_ptr() % 16 == 0:
                        from ._cabi import call
                        from .ops import _stream
                        call("miseg_copy", _stream(), slot.data_ptr(), g.data_ptr(), slot.numel() * 4)   # a kernel, not hipMemcpyDtoD (which can stall for 100s of us on a busy device)
                    elif slot.is_cuda and g.dtype == slot.dtype and g.shape == slot.shape:
                        torch.mul(g, 1.0, out=slot)
                    else:
                        slot.copy_(g)
                    if g.is_cuda:
                        from .tape import keep
                        keep(g, torch.cuda.current_stream(g.device))   # may be another stream than g's own (GradReducer._launch)
                else:
                    continue
                p.grad = slot


class LossScaler:
    """Dynamic loss scale of the half-precision storage mode (BASELINE configs[4]), torch.cuda.amp.GradScaler's policy without its
    host synchronisation: the device counts the non-finite entries of the flat gradient (``miseg_count_nonfinite``), that count is
    one more guard flag of the fused Adam launch -- an overflowed gradient moves neither weights nor moments -- and it reaches the
    host with the iteration's scalars, one iteration late, where ``update`` halves the scale (down to 1) or, after
    ``growth_interval`` clean iterations in a row, doubles it (up to ``max_scale``)."""

    def __init__(self, init_scale: float, growth_interval: int = 2000, max_scale: float = 65536.0):
        self.scale, self.growth_interval, self.max_scale = float(init_scale), int(growth_interval), float(max_scale)
        self.good, self.overflows = 0, 0

    def update(self, nonfinite: float, used_scale: Optional[float] = None) -> None:
        """``nonfinite``: the iteration's count of inf / NaN gradient entries; ``used_scale``: the scale that iteration was staged with.
        The count arrives one iteration late, so the iteration after an overflow has already been staged with the same scale and
        overflows too: only an overflow at the CURRENT scale halves it (one overflow event = one halving, as torch.cuda.amp.GradScaler
        does with its synchronous check); every overflowed iteration is counted and has skipped its update on the device."""
        if nonfinite != 0.0 or nonfinite != nonfinite:
            self.overflows += 1
            self.good = 0
            if used_scale is None or used_scale <= self.scale:
                self.scale = max(1.0, self.scale * 0.5)
        else:
            self.good += 1
            if self.good >= self.growth_interval:
                self.good = 0
                self.scale = min(self.max_scale, self.scale * 2.0)

    def state_dict(self) -> dict:
        return {"scale": self.scale, "good": self.good, "overflows": self.overflows}

    def load_state_dict(self, sd: dict) -> None:
        self.scale, self.good, self.overflows = float(sd["scale"]), int(sd["good"]), int(sd.get("overflows", 0))


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (amsgrad=False) as one fused HIP launch over flat buffers."""

    def __init__(self, params: Iterable, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not implemented in the fused HIP Adam")
        # the remaining keys are torch.optim.Adam's own defaults: carried so that param_groups -- hence checkpoints -- have the
        # key set a torch Adam of this torch version writes (reference checkpoints load key for key, and ours load there)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        self._flats: List[FlatBuffers] = [FlatBuffers(list(g["params"])) for g in self.param_groups]
        self._steps = [0 for _ in self.param_groups]
        self._m: List[Optional[Tensor]] = [None] * len(self.param_groups)
        self._v: List[Optional[Tensor]] = [None] * len(self.param_groups)
        self._hyper: List[Optional[Tensor]] = [None] * len(self.param_groups)
        self._hyper_host: List[Optional[Tensor]] = []
        self._pending_state: Optional[dict] = None
        self.grad_scale = 1.0   # gradients arrive multiplied by this (loss scale of the fp16 mode); divided out in the kernel
        self.loss_scaler: Optional[LossScaler] = None      # set by the train epocher in the fp16 storage mode
        self.last_nonfinite: Optional[Tensor] = None       # device float[1] of the latest ``apply`` in that mode

    @property
    def flat(self) -> FlatBuffers:
        return self._flats[0]

    def zero_grad(self, set_to_none: bool = False) -> None:
        for fb in self._flats:
            fb.zero_grad()

    def _ensure_state(self, gi: int) -> None:
        fb = self._flats[gi]
        rebuilt = not fb.valid()
        fb.ensure()
        if self._m[gi] is None or rebuilt and self._m[gi].numel() != fb.total or self._m[gi].device != fb.flat_param.device:
            self._m[gi] = torch.zeros_like(fb.flat_param)
            self._v[gi] = torch.zeros_like(fb.flat_param)
            self._hyper[gi] = torch.zeros(4, dtype=torch.float32, device=fb.flat_param.device)
        if self._pending_state is not None:
            self._apply_state(self._pending_state)
            self._pending_state = None

    @torch.no_grad()
    def host_step(self) -> List[List[float]]:
        """Host half of a step without any device traffic: bump the step counters and return, per parameter group, the four scalars
        of the update (lr / bias correction 1, 1 / sqrt(bias correction 2), eps, weight decay).  The training loop stages them in the
        iteration's step block (``stepio.StepIO.begin``), whose single upload carries them; ``apply(io=...)`` reads them there."""
        rows = []
        for gi, group in enumerate(self.param_groups):
            self._ensure_state(gi)
            self._steps[gi] += 1
            t = self._steps[gi]
            b1, b2 = group["betas"]
            bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
            rows.append([group["lr"] / bc1, 1.0 / math.sqrt(bc2), group["eps"], group["weight_decay"]])
        return rows

    @torch.no_grad()
    def advance(self) -> None:
        """Host half of a step: bump the step counters and upload (lr / bias corrections, eps, weight decay) to the device
        hyper-parameter array.  Kept apart from ``apply`` so that a captured hipGraph of the step (semi_seg.epocher) can
        replay the device half while the host half runs before every replay."""
        for gi, group in enumerate(self.param_groups):
            self._ensure_state(gi)
            self._steps[gi] += 1
            t = self._steps[gi]
            b1, b2 = group["betas"]
            bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
            # Pinned staging, FOUR slots used round-robin: the copy below is asynchronous and the training loop reads iteration i's
            # scalars only after iteration i+1 is enqueued (semi_seg/epocher.py _after_step), so the host can be two steps ahead of
            # the copy engine -- a single slot would be rewritten for step i+2 while step i+1's copy may still be queued.
            while len(self._hyper_host) <= gi:
                self._hyper_host.append(None)
            vals = torch.tensor([group["lr"] / bc1, 1.0 / math.sqrt(bc2), group["eps"], group["weight_decay"]], dtype=torch.float32)
            if not self._hyper[gi].is_cuda:
                self._hyper[gi].copy_(vals)
                continue
            ring = self._hyper_host[gi]
            if ring is None:
                from .ops import PinnedRing
                ring = self._hyper_host[gi] = PinnedRing((4,), torch.float32, slots=4)
            ring.upload_into(lambda slot: slot.copy_(vals), self._hyper[gi])

    @torch.no_grad()
    def apply(self, guard: Optional[Tensor] = None, io=None) -> None:
        """Device half: gather stray gradients into the flat buffer and launch the fused Adam kernel (a no-op on the device if any
        of the fp32 flags in ``guard`` is set).  ``io`` = the iteration's step block (``stepio.StepIO``): the update's scalars -- and
        the loss scale of the half-precision mode -- are read from it on the device, and the overflow count lands in its read-back
        block right behind the guard flags (one contiguous guard vector, no ``torch.cat``)."""
        self.last_nonfinite = None
        if io is not None and (len(self.param_groups) == 1 or self.loss_scaler is None):
            for gi, group in enumerate(self.param_groups):
                fb = self._flats[gi]
                b1, b2 = group["betas"]
                fb.collect()
                scale = 1.0
                if self.loss_scaler is not None:
                    bad = io.overflow_slot(guard)          # fp32[1] behind the flags (or on its own when nothing guards)
                    unet_ops.count_nonfinite(fb.flat_grad, out=bad)
                    self.last_nonfinite = bad
                    guard = io.guard_with_overflow(guard, bad)
                    scale = -1.0                           # 1 / loss scale is hyper[4] of the step block
                unet_ops.adam_step(fb.flat_param, fb.flat_grad, self._m[gi], self._v[gi], io.hyper(gi), b1, b2, scale, guard)
            unet_ops.PACK_CACHE.invalidate()
            return
        for gi, group in enumerate(self.param_groups):
            fb = self._flats[gi]
            b1, b2 = group["betas"]
            fb.collect()
            if self.loss_scaler is not None and fb.flat_grad.is_cuda:
                # half-precision storage: a gradient that left half's range is inf / NaN here -- count, and let the count guard the update
                bad = unet_ops.count_nonfinite(fb.flat_grad)
                self.last_nonfinite = bad if self.last_nonfinite is None else self.last_nonfinite + bad
                guard = bad if guard is None else torch.cat([guard.reshape(-1), bad])
            unet_ops.adam_step(fb.flat_param, fb.flat_grad, self._m[gi], self._v[gi], self._hyper[gi], b1, b2, self.grad_scale, guard)
        unet_ops.PACK_CACHE.invalidate()   # the fp32 masters changed: packed operand copies are stale

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.advance()
        self.apply()
        return loss

    # ---- torch.optim.Adam-compatible checkpoints (per-parameter exp_avg / exp_avg_sq / step)
    def step_skipped(self) -> None:
        """An iteration reported (one iteration late) that the device skipped its update (gradient overflow in the fp16 mode): take it
        back out of the step counters, so that the bias corrections count applied updates only, as torch.optim.Adam under a GradScaler."""
        self._steps = [max(0, t - 1) for t in self._steps]

    def state_dict(self) -> dict:
        state: Dict[int, dict] = {}
        groups = []
        idx = 0
        for gi, group in enumerate(self.param_groups):
            fb = self._flats[gi]
            ids = []
            for p in group["params"]:            # torch convention: ids follow the optimiser's parameter order
                if self._m[gi] is not None and fb.offsets:
                    o = fb.offset_of(p)
                    state[idx] = {"step": torch.tensor(float(self._steps[gi])),
                                  "exp_avg": self._m[gi][o:o + p.numel()].view_as(p).clone(),
                                  "exp_avg_sq": self._v[gi][o:o + p.numel()].view_as(p).clone()}
                ids.append(idx)
                idx += 1
            groups.append({**{k: v for k, v in group.items() if k != "params"}, "params": ids})
        out = {"state": state, "param_groups": groups}
        if self.loss_scaler is not None:       # fp16 storage mode only (the fp32 / bf16 checkpoints keep torch.optim.Adam's exact key set)
            out["loss_scaler"] = self.loss_scaler.state_dict()
        return out

    def _apply_state(self, sd: dict) -> None:
        idx = 0
        for gi, group in enumerate(self.param_groups):
            fb = self._flats[gi]
            for p in group["params"]:
                st = sd["state"].get(idx, sd["state"].get(str(idx)))
                if st is not None:
                    o = fb.offset_of(p)
                    self._m[gi][o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                    self._v[gi][o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                    self._steps[gi] = int(float(st["step"]))
                idx += 1

    def load_state_dict(self, state_dict: dict) -> None:
        if state_dict.get("loss_scaler") is not None:
            if self.loss_scaler is None:
                self.loss_scaler = LossScaler(float(state_dict["loss_scaler"]["scale"]))
            self.loss_scaler.load_state_dict(state_dict["loss_scaler"])
        for group, saved in zip(self.param_groups, state_dict["param_groups"]):
            for k, v in saved.items():
                if k != "params":
                    group[k] = v
        if all(fb.params[0].is_cuda for fb in self._flats):
            for gi in range(len(self.param_groups)):
                self._ensure_state(gi)
            self._apply_state(state_dict)
        else:
            self._pending_state = state_dict  # applied once the parameters live on the GPU
