"""Host <-> device traffic of one training iteration, in two copies.

The reference's loop moves small things between host and device all the time: the per-sample flip decisions (ref
semi_seg/epocher.py:148-149), the optimiser's scalars, ~7 ``.item()`` read-backs (ref :181-185, :225, :278-282).  Here everything the
host sends per iteration -- flip masks, Adam's step scalars, the loss coefficients that seed backward (times the loss scale of the
fp16 mode), zeroed assertion counters -- lives in ONE device block refreshed by ONE ``miseg_upload`` from a pinned ring, and everything
it reads back -- meter values, deferred assertion flags, Dice counts, the overflow count -- in ONE device block fetched by ONE
``miseg_download``.  Both are library entry points, so they are on the launch tape (``miseg_amd.tape``) like the kernels between them, and
the kernels write their scalar results straight into these blocks (no ``torch.cat`` / ``torch.zeros`` / ``fill_`` launches around them).

A ``StepIO`` is installed as ``stepio.CURRENT`` for the duration of an iteration; ops ask it for ``counter()`` / ``scalar()`` /
``seed()`` views and fall back to ordinary torch tensors when none is installed (stand-alone use of an op, the tests of single kernels).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _cabi
from ._cabi import call

FLIPS_OFF, FLIPS_CAP = 0, 512                 # int32
HYPER_OFF, HYPER_CAP = 2048, 64               # float32: 8 per parameter group (lr/bc1, 1/sqrt(bc2), eps, wd, 1/loss scale)
COUNT_OFF, COUNT_CAP = 2304, 64               # int32, zero on the host: the upload zeroes the counters
import os as _os
# (the opt-in accumulators of the BatchNorm backward need 6 892 more words)
SEED_OFF, ACC_OFF, PARAM_BYTES = 2560, 32768, 131072 if _os.environ.get("MISEG_BN_ACC_BWD", "0") == "1" else 65536
SEED_CAP = (ACC_OFF - SEED_OFF) // 4
ACC_CAP = (PARAM_BYTES - ACC_OFF) // 8        # int64, zero on the host: the BatchNorm statistics accumulators (miseg_conv3x3_fwd_acc)
ARENA_FLOATS = 8192
OUT_BYTES = 65536

CURRENT: Optional["StepIO"] = None


def current() -> Optional["StepIO"]:
    return CURRENT


class Ticket:
    """What ``StepIO.post`` hands back: where the iteration's read-back will land and the event that says it has."""
    __slots__ = ("io", "slot", "fields", "names", "items", "nvals", "scale")

    def __init__(self, io, slot, fields, names, items, nvals):
        self.io, self.slot, self.fields, self.names, self.items, self.nvals = io, slot, fields, names, items, nvals
        self.scale = io.scale          # the loss scale this iteration was staged with (fp16 mode: flat.LossScaler.update)


class StepIO:
    def __init__(self, device, slots: int = 4):
        self.device = torch.device(device)
        lib = _cabi.lib()
        self.dev = torch.zeros(PARAM_BYTES, dtype=torch.uint8, device=self.device)
        self.host = torch.zeros(slots, PARAM_BYTES, dtype=torch.uint8).pin_memory()
        self.up_events = [lib.miseg_event_create() for _ in range(slots)]
        self.up_pending = [False] * slots
        self.turn = 0
        self.arena = torch.zeros(ARENA_FLOATS, dtype=torch.float32, device=self.device)
        self.out_dev = torch.zeros(OUT_BYTES, dtype=torch.uint8, device=self.device)
        self.out_host = torch.zeros(slots, OUT_BYTES, dtype=torch.uint8).pin_memory()
        self.out_events = [lib.miseg_event_create() for _ in range(slots)]
        self.out_turn = 0
        self._seeds: Dict[Tuple[int, float], Tuple[int, int]] = {}     # (numel, coefficient) -> (offset in floats, numel)
        self._seed_used = 0
        self._n_flips = 0
        self._groups = 0
        self.scale = 1.0
        self._cursor_counter = self._cursor_arena = self._cursor_out = self._cursor_acc = 0
        self._fields: List[Tuple[str, int, int, torch.dtype, tuple]] = []
        # typed views of the device block
        self._d_i32 = self.dev.view(torch.int32)
        self._d_f32 = self.dev.view(torch.float32)
        self._d_i64 = self.dev.view(torch.int64)

    def __del__(self):
        try:
            lib = _cabi.lib()
            for ev in self.up_events + self.out_events:
                lib.miseg_event_destroy(ev)
        except Exception:
            pass

    # ------------------------------------------------------------------ host -> device
    def stage(self, flip_masks: Sequence[int], hyper_rows: Sequence[Sequence[float]], scale: float = 1.0) -> int:
        """Host half of ``begin``: write this iteration's values into the next pinned slot (waiting, if need be, for the copy that
        last read it) and reset the per-iteration cursors; returns the slot.  A replayed launch tape calls this alone -- the upload
        is on the tape, its source pointer bound to the slot."""
        n = len(flip_masks)
        if n > FLIPS_CAP or len(hyper_rows) * 8 > HYPER_CAP:
            raise _cabi.MisegError(f"StepIO: {n} flip masks / {len(hyper_rows)} parameter groups exceed the block layout")
        self.turn = (self.turn + 1) % len(self.up_events)
        slot = self.turn
        if self.up_pending[slot]:
            call("miseg_event_synchronize", self.up_events[slot])     # free unless the host is >= `slots` uploads ahead
        h = self.host[slot]
        hi, hf = h.view(torch.int32), h.view(torch.float32)
        if n:
            hi[FLIPS_OFF // 4:FLIPS_OFF // 4 + n] = torch.as_tensor(list(flip_masks), dtype=torch.int32)
        self._n_flips, self._groups, self.scale = n, len(hyper_rows), float(scale)
        for gi, row in enumerate(hyper_rows):
            vals = list(row)[:4] + [1.0 / float(scale)]
            hf[HYPER_OFF // 4 + 8 * gi:HYPER_OFF // 4 + 8 * gi + 5] = torch.tensor(vals, dtype=torch.float32)
        for (numel, coeff), (off, _) in self._seeds.items():
            hf[SEED_OFF // 4 + off:SEED_OFF // 4 + off + numel] = coeff * float(scale)
        self._cursor_counter = self._cursor_arena = self._cursor_out = self._cursor_acc = 0
        self._fields = []
        self.last_report = None
        return slot

    def upload(self, slot: int) -> None:
        """The one host -> device copy of the iteration (current stream) and the event that frees the slot."""
        from .ops import _stream
        st = _stream()
        call("miseg_upload", st, self.dev.data_ptr(), self.host[slot].data_ptr(), PARAM_BYTES)
        call("miseg_event_record", st, self.up_events[slot])
        self.up_pending[slot] = True

    def begin(self, flip_masks: Sequence[int], hyper_rows: Sequence[Sequence[float]], scale: float = 1.0) -> None:
        self.upload(self.stage(flip_masks, hyper_rows, scale))

    def flips(self, n: Optional[int] = None) -> Tensor:
        n = self._n_flips if n is None else n
        return self._d_i32[FLIPS_OFF // 4:FLIPS_OFF // 4 + n]

    def hyper(self, gi: int) -> Tensor:
        return self._d_f32[HYPER_OFF // 4 + 8 * gi:HYPER_OFF // 4 + 8 * gi + 5]

    def counter(self) -> Tensor:
        """A 0-d int32 device counter that is zero at the start of the iteration (the upload wrote it)."""
        i = self._cursor_counter
        if i >= COUNT_CAP:
            raise _cabi.MisegError("StepIO: out of assertion counters")
        self._cursor_counter += 1
        return self._d_i32[COUNT_OFF // 4 + i]

    def acc64(self, n: int) -> Optional[Tensor]:
        """``n`` int64 device words that are zero at the start of the iteration (the upload wrote them): a fixed-point statistics
        accumulator of one BatchNorm layer.  None when the block has no room left (the caller zero-fills a tensor of its own)."""
        i = self._cursor_acc
        if i + n > ACC_CAP:
            return None
        self._cursor_acc += n
        return self._d_i64[ACC_OFF // 8 + i:ACC_OFF // 8 + i + n]

    def counters_base(self) -> Tensor:
        return self._d_i32[COUNT_OFF // 4:COUNT_OFF // 4 + COUNT_CAP]

    def seed(self, shape, coeff: float) -> Tensor:
        """fp32 device tensor of ``shape`` filled with ``coeff * loss scale``: the constant gradient that seeds backward at a loss
        kernel's output.  Registered once; from then on every ``begin`` writes it with the iteration's loss scale."""
        numel = 1
        for d in shape:
            numel *= int(d)
        key = (numel, float(coeff))
        ent = self._seeds.get(key)
        if ent is None:
            if self._seed_used + numel > SEED_CAP:
                raise _cabi.MisegError("StepIO: out of seed space")
            ent = self._seeds[key] = (self._seed_used, numel)
            self._seed_used += (numel + 3) // 4 * 4
            off = SEED_OFF // 4 + ent[0]
            # first use (a warm-up iteration): fill the device copy now and every pinned slot, so that whichever slot goes up next has it
            # (through .data: an alias with its own version counter -- the block's views are saved by autograd nodes of this iteration)
            self.dev.data.view(torch.float32)[off:off + numel].fill_(float(coeff) * self.scale)
            self.host.view(torch.float32).view(len(self.up_events), -1)[:, off:off + numel] = float(coeff) * self.scale
        off = SEED_OFF // 4 + ent[0]
        return self._d_f32[off:off + numel].view(tuple(int(d) for d in shape))

    # ------------------------------------------------------------------ scalar results of kernels
    def scalar(self, shape=()) -> Tensor:
        """fp32 device tensor for a kernel's small result (a loss, a vector of per-head losses), bump-allocated from one arena
        so that the iteration's report reads them in place."""
        numel = 1
        for d in shape:
            numel *= int(d)
        i = self._cursor_arena
        if i + numel > ARENA_FLOATS:
            raise _cabi.MisegError("StepIO: scalar arena exhausted")
        self._cursor_arena += numel
        return self.arena[i:i + numel].view(tuple(int(d) for d in shape))

    def arena_offset(self, t: Tensor) -> Optional[int]:
        """Index of ``t``'s first element in the arena, or None if it does not live there."""
        if t.dtype != torch.float32 or not t.is_contiguous():
            return None
        d = t.data_ptr() - self.arena.data_ptr()
        if d < 0 or d % 4 or d // 4 + t.numel() > self._cursor_arena:
            return None
        return d // 4

    def counter_index(self, t: Tensor) -> Optional[int]:
        if t.dtype != torch.int32:
            return None
        d = t.data_ptr() - self.counters_base().data_ptr()
        if d < 0 or d % 4 or d // 4 >= self._cursor_counter:
            return None
        return d // 4

    # ------------------------------------------------------------------ device -> host
    def out(self, name: str, shape, dtype) -> Tensor:
        """Device tensor inside the read-back block; comes back under ``name`` in the ticket's host fields."""
        numel = 1
        for d in shape:
            numel *= int(d)
        es = torch.empty((), dtype=dtype).element_size()
        off = (self._cursor_out + 15) // 16 * 16
        if off + numel * es > OUT_BYTES:
            raise _cabi.MisegError("StepIO: read-back block exhausted")
        self._cursor_out = off + numel * es
        shape = tuple(int(d) for d in shape)
        self._fields.append((name, off, numel, dtype, shape))
        return self.out_dev[off:off + numel * es].view(dtype).view(shape)

    def overflow_slot(self, guard: Optional[Tensor]) -> Tensor:
        """fp32[1] for the half-precision mode's overflow count: the free slot ``lazy._report_in_place`` left behind the flags when
        ``guard`` is that report's flag vector, else a field of its own."""
        rep = getattr(self, "last_report", None)
        if guard is not None and rep is not None and guard.numel() and guard.data_ptr() + 4 * guard.numel() == rep.data_ptr() + 4 * (rep.numel() - 1):
            return rep[-1:]
        return self.out("nonfinite", (1,), torch.float32)

    @staticmethod
    def guard_with_overflow(guard: Optional[Tensor], bad: Tensor) -> Tensor:
        if guard is None or not guard.numel():
            return bad
        if guard.data_ptr() + 4 * guard.numel() == bad.data_ptr():
            return guard.as_strided((guard.numel() + 1,), (1,), guard.storage_offset())
        return torch.cat([guard.reshape(-1), bad])

    def post(self, names: List[str], items: list, nvals: int) -> Ticket:
        """Enqueue the one download of the block and the event behind it (current stream)."""
        self.out_turn = (self.out_turn + 1) % len(self.out_events)
        slot = self.out_turn
        from .ops import _stream
        st = _stream()
        nbytes = (self._cursor_out + 15) // 16 * 16
        call("miseg_download", st, self.out_host[slot].data_ptr(), self.out_dev.data_ptr(), nbytes)
        call("miseg_event_record", st, self.out_events[slot])
        return Ticket(self, slot, list(self._fields), list(names), list(items), nvals)

    def wait(self, ticket: Ticket) -> Dict[str, Tensor]:
        """Host views of the ticket's fields (valid until ``slots - 1`` further posts)."""
        call("miseg_event_synchronize", self.out_events[ticket.slot])
        h = self.out_host[ticket.slot]
        return {name: h[off:off + numel * torch.empty((), dtype=dt).element_size()].view(dt).view(shape)
                for name, off, numel, dt, shape in ticket.fields}
