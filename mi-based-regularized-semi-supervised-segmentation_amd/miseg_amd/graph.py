"""One training iteration as a replayed hipGraph.

A udaiic step issues ~560 kernel launches; through Python + ctypes that is ~12.6 ms of host time per step at the
BASELINE cfg2 shape -- as long as the GPU work itself (13 ms).  The device half of the iteration
(``TrainEpocher._device_step``: forward, losses, backward, gradient gather, fused Adam, Dice counts) has no host
synchronisation and no host->device traffic, so it is captured once (``torch.cuda.graph`` = hipStreamBeginCapture on
the stream all our launches go to) and replayed; per iteration the host only
  * draws the flip decisions and copies the batches / flip masks into the graph's static input buffers,
  * runs ``optimizer.advance()`` (step counter, lr and bias corrections -> the device hyper-parameter array),
  * replays, then reads the static output scalars with the iteration's single blocking fetch.
Shapes are fixed per graph; a batch of another shape (e.g. a short last batch) runs eagerly.

Stream discipline (the PyTorch capture recipe, and measured here): every iteration this object runs -- the eager warm-up
ones too -- executes on ONE dedicated side stream, and the capture uses that stream.  Capturing a backward pass after an
eager backward has run on the legacy default stream in the same process crashed inside the HIP runtime at
hipStreamEndCapture; with the warm-up on the capture stream it does not.  Enable the graph before the first training
iteration of the process (evaluation / forward-only work on the default stream is harmless).
"""
from __future__ import annotations

from typing import List

import torch
from torch import Tensor

from . import unet_ops


class StepGraph:
    def __init__(self, epocher, warmup: int = 3):
        self.ep = epocher
        self.warmup = warmup          # eager iterations before capture (allocator pool, cached constants, pack registry)
        self.seen = 0
        self.graph = None
        self.key = None
        self.stream = None

    def _eager(self, li, lt, ui, masks, seed):
        flips2 = torch.tensor(masks, dtype=torch.int32, device=li.device)
        self.ep._optimizer.advance()
        return self.ep._device_step(li, lt, ui, flips2, seed)

    def run(self, li: Tensor, lt: Tensor, ui: Tensor, masks: List[int], seed: int):
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=li.device)
        caller = torch.cuda.current_stream(li.device)
        self.stream.wait_stream(caller)
        with torch.cuda.stream(self.stream):
            out = self._run(li, lt, ui, masks, seed)
        caller.wait_stream(self.stream)
        return out

    def _run(self, li: Tensor, lt: Tensor, ui: Tensor, masks: List[int], seed: int):
        ep = self.ep
        key = (tuple(li.shape), tuple(lt.shape), tuple(ui.shape), li.dtype, lt.dtype, len(masks))
        if self.graph is None and self.seen < self.warmup:
            self.seen += 1
            return self._eager(li, lt, ui, masks, seed)
        if self.graph is not None and key != self.key:
            return self._eager(li, lt, ui, masks, seed)
        if self.graph is None:
            self._capture(li, lt, ui, masks, seed, key)
        self.s_li.copy_(li, non_blocking=True)
        self.s_lt.copy_(lt, non_blocking=True)
        self.s_ui.copy_(ui, non_blocking=True)
        # pinned staging ring (the host may run two steps ahead of the copy engine; a slot waits for its own last copy before reuse)
        self.h_masks.upload_into(lambda slot: slot.copy_(torch.tensor(masks, dtype=torch.int32)), self.s_masks)
        ep._optimizer.advance()
        self.graph.replay()
        unet_ops.PACK_CACHE.invalidate()       # the replay ran Adam: eager users (evaluation) must re-pack
        ep._pending.set_static(self.names, self.s_scalars, self.items)
        return self.s_inter, self.s_union

    def _capture(self, li, lt, ui, masks, seed, key):
        ep = self.ep
        dev = li.device
        self.s_li, self.s_lt, self.s_ui = li.clone(), lt.clone(), ui.clone()
        self.s_masks = torch.zeros(len(masks), dtype=torch.int32, device=dev)
        from .ops import PinnedRing
        self.h_masks = PinnedRing((len(masks),), torch.int32, slots=4)
        ep._pending.drain()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=self.stream):
            inter, union = ep._device_step(self.s_li, self.s_lt, self.s_ui, self.s_masks, seed)
            ep._pending.precompute()       # (already done by the guarded optimiser launch unless MISEG_GUARD_STEP=0)
            names, self.s_scalars, items = ep._pending.take_static()
        self.names, self.items = names, items
        self.s_inter, self.s_union = inter, union
        self.graph, self.key = graph, key
