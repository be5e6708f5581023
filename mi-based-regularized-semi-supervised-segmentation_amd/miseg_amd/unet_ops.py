"""Autograd front-ends of the U-Net kernels: conv3x3 + BatchNorm(train/eval) + ReLU (+ fused 2x2
max-pool, + fused nearest-x2 upsample / skip concat on the input side) and the 1x1 logits head.

ref: contrastyou/arch/unet.py:10-40 (conv_block / up_conv), :61-64 (pools), :84,129 (DeConv_1x1),
:109-125 (torch.cat skips).  Activations are [N,C,H,W]-shaped torch tensors in channels_last memory
(= the NHWC layout of the kernels) of dtype float32 (exact mode) or bfloat16.
"""
from __future__ import annotations

import os

from typing import Optional, Tuple

import struct
import weakref

import torch
from torch import Tensor

from ._cabi import call, query
from .gradslot import grad_slot
from .ops import _DT, _need_gpu, _ptr, _stream, _ws, as_nhwc, empty_nhwc, wait_stream
from .tape import keep
from . import stepio

BN_EPS, BN_MOMENTUM = 1e-5, 0.1

# ---- weight gradients off the critical path
# In the U-Net backward only BN-backward -> dgrad feeds the next layer; a layer's wgrad (+ its split reduction) has no consumer
# before the optimiser.  It is launched on a side stream so the dgrad chain does not queue behind it and the two fill each
# other's tails.  Joined before anything reads the flat gradient buffer (FlatBuffers.collect, GradReducer._launch).
_WGRAD_SIDE = os.environ.get("MISEG_WGRAD_STREAM", "1") != "0"
_GRAPH_STREAMS = os.environ.get("MISEG_GRAPH_STREAMS", "1") == "1"   # fork side streams inside a captured step too
_wgrad_streams: dict = {}
_wgrad_dirty: set = set()


def wgrad_stream(device):
    st = _wgrad_streams.get(device)
    if st is None:
        st = _wgrad_streams[device] = torch.cuda.Stream(device=device)
    return st


def join_wgrad_streams() -> None:
    """Make the current stream wait for every outstanding side-stream wgrad.  Called with the weight-gradient stream itself current (the
    data-parallel reducer launches its collectives from it) there is nothing to wait for and the device stays marked: the backward
    pass's own stream still has to join when the pass ends."""
    for dev in list(_wgrad_dirty):
        cur = torch.cuda.current_stream(dev)
        if cur != _wgrad_streams[dev]:
            wait_stream(cur, _wgrad_streams[dev])
            _wgrad_dirty.discard(dev)


_FUSE_UPS_DGRAD = True   # False: conv3x3 dgrad + miseg_sumpool2x2 as two launches
_DUAL_DGRAD = True       # concat layers: one data-gradient launch with two destinations (False: one launch per source)
# BatchNorm backward folded into the convolutions around it (layers without the fused pool): the statistics pass leaves six
# coefficients per channel, the data- and weight-gradient kernels form graw in their loaders (no bn_bwd_apply pass, no graw tensor),
# and a data-gradient launch whose output is the activation gradient of ANOTHER such layer takes that layer's statistics pass in its
# epilogue (no bn_relu_bwd_reduce pass there).  OFF by default (MISEG_BN_FUSE=1 turns it on): measured slower, DESIGN.md section 9.
_FUSE_BN_BWD = os.environ.get("MISEG_BN_FUSE", "0") == "1"
_FUSE_BN_RED = True      # ... including the statistics pass in the producing data-gradient kernel's epilogue
# The statistics pass ALONE in the producing data-gradient kernel's epilogue (no loader transform): MISEG_BN_RED=1 (round-4 experiment)
_RED_ONLY = os.environ.get("MISEG_BN_RED", "0") == "1" and not _FUSE_BN_BWD
# 1 (default): the forward statistics leave the convolution as fixed-point atomic adds and the apply kernel finishes them itself
# (miseg_conv3x3_fwd_acc / miseg_bn_relu_fwd_acc): 22 launches fewer per step.  0: one partial row per block + miseg_bn_finalize.
# The stem (one image channel) through its own pixel-per-thread kernels reading the fp32 image, instead of the streaming MFMA convolution
# and the tiled weight gradient on a padded channel vector: the kernels themselves are on a par (50 vs 50 us forward, 52 + 7 vs 59 + 10 us
# weight gradient -- the step's last kernel), the padded operand's 15 us launch becomes 4: -0.02 ms per step.  0: the MFMA path.
_STEM_KERNELS = os.environ.get("MISEG_STEM_KERNELS", "1") != "0"
_STEM_MAX_W = 1022                  # the stem kernels keep whole image rows (+ halo) in 4 x 256 registers
_BN_ACC = os.environ.get("MISEG_BN_ACC", "1") != "0"
# The same for the backward's sums (two fixed-point tiers, miseg_bn_relu_bwd_dual_acc).  Built and measured, OFF: the reduce kernel's
# blocks all finish together, so their atomics arrive as one burst on a few lines (+8 us per launch) and the apply prologue costs 5 us
# -- more than the finalize launch they replace (6 us + a launch boundary).  DESIGN.md section 10.
_BN_ACC_BWD = os.environ.get("MISEG_BN_ACC_BWD", "0") == "1"


class _BnRec:
    """What a layer's CONSUMER needs to take the layer's BatchNorm-backward sums in the epilogue of its data-gradient kernel, and
    the hand-over of those sums.  Hangs off the activation tensor (``y._miseg_bn``); the consumer's backward fills ``parts`` and
    names the gradient tensor they belong to, the layer's backward uses them only if autograd hands it that very tensor, unmodified
    (another consumer of the activation -> autograd added the gradients into a new tensor -> the ordinary reduce runs)."""
    __slots__ = ("raw", "saved", "shape", "parts", "nparts", "for_grad", "for_version")

    def __init__(self):
        self.raw = self.saved = self.shape = self.parts = self.for_grad = None
        self.nparts = self.for_version = 0

    def offer(self, grad: Tensor, parts: Tensor, nparts: int) -> None:
        self.parts, self.nparts, self.for_grad, self.for_version = parts, nparts, grad, grad._version

    def take(self, grad: Optional[Tensor]):
        parts, nparts, mine = self.parts, self.nparts, self.for_grad is grad and grad is not None and grad._version == self.for_version
        self.parts = self.for_grad = None
        return (parts, nparts) if mine and parts is not None else (None, 0)


class _GradSink:
    """Side channel for a SECOND gradient of a layer's activation.  A tapped feature map feeds the next layer and a cluster head (ref
    semi_seg/epocher.py:258-273, the head sees the last 2 UB samples); autograd would add the two gradients with an elementwise kernel
    over the whole batch, behind a zero fill of the samples the head does not touch.  Instead the head's backward ``put``s its compact
    gradient (samples [n0, n1) only) here and returns no gradient; this layer's backward -- which autograd runs after every consumer
    of the activation, whatever they return -- ``take``s it and the BatchNorm-backward kernels add it in their loaders
    (``miseg_bn_relu_bwd_dual``)."""
    __slots__ = ("items",)

    def __init__(self):
        self.items = []

    def put(self, grad: Tensor, n0: int, n1: int) -> None:
        self.items.append((grad, int(n0), int(n1), torch.cuda.current_stream(grad.device)))

    def take(self):
        items, self.items = self.items, []
        return items


class _SyncCounters:
    """int32 counters for the "last block finishes" launches (miseg_conv3x3_bn_fwd, miseg_bn_relu_bwd_sync): zero when a launch
    starts, zero again when it ends.  One zero-filled array per device, handed out round-robin, so that launches which could overlap
    (different streams) never share a counter.
    OFF by default (``MISEG_BN_FINISH=1`` turns it on): it removes 44 of the step's ~300 launches, but the finishing block's serial
    tail (ticket atomic -> cache invalidate -> dependent loads of the partial rows from memory -> coefficients) costs more than the
    5 us finalize kernel plus its launch gap did: 8.54 ms/step (rows written through, no release fence) and 8.83 ms (release fence
    per block) against 8.03-8.10 ms with the separate finalize launches, same box, same run."""
    SLOTS = 256

    def __init__(self):
        self._pool, self._next = {}, 0
        self.enabled = os.environ.get("MISEG_BN_FINISH", "0") == "1"

    def take(self, device) -> Optional[Tensor]:
        if not self.enabled:
            return None
        pool = self._pool.get(device)
        if pool is None:
            pool = self._pool[device] = torch.zeros(self.SLOTS, dtype=torch.int32, device=device)
        self._next = (self._next + 1) % self.SLOTS
        return pool[self._next:self._next + 1]


SYNC_COUNTERS = _SyncCounters()


def loss_scale_of(model) -> float:
    """INITIAL loss scale of a model: 1 unless its activations are stored as IEEE half, whose range (6e-8 .. 65504) does not hold the
    per-pixel gradients of a mean over 48 x 256 x 256 positions (~3e-7).  ``MISEG_LOSS_SCALE`` overrides the default 2^14; the
    fused Adam divides it back out (``miseg_adam_step_scaled``), so the update equals the unscaled one up to rounding.  From there the
    scale is dynamic (``flat.LossScaler``: halved when the flat gradient holds an inf / NaN, that step skipped on the device)."""
    net = getattr(model, "module", model)
    if getattr(net, "compute_dtype", None) != torch.float16:
        return 1.0
    return float(os.environ.get("MISEG_LOSS_SCALE", 16384.0))


def vec_of(dtype) -> int:
    return 8 if dtype in (torch.bfloat16, torch.float16) else 4


def stem_input(image: Tensor, dtype) -> Tensor:
    """fp32 [B,Cin,H,W] image -> [B,VEC,H,W] channels_last tensor of ``dtype`` (extra channels zero)."""
    _need_gpu(image)
    pre = getattr(image, "_miseg_stem", None)      # the launch that assembled the batch wrote this operand too (ops.cat_flip)
    if pre is not None and pre[0] == dtype and pre[1].shape[0] == image.shape[0]:
        return pre[1]
    b, cin, h, w = image.shape
    if _STEM_KERNELS and cin == 1 and w <= _STEM_MAX_W and image.dtype == torch.float32 and image.is_contiguous() and dtype in (torch.bfloat16, torch.float16):
        # the stem's own kernels read the fp32 image itself (and round it as they read): the padded operand is only DESCRIBED -- an
        # uninitialised tensor of its shape and type that carries the image; conv_bn_relu fills it in if it takes the MFMA path after all
        out = empty_nhwc(b, vec_of(dtype), h, w, dtype, image.device)
        out._miseg_stem_f32 = image
        return out
    image = as_nhwc(image.float())
    cp = vec_of(dtype)
    cp = ((cin + cp - 1) // cp) * cp
    out = empty_nhwc(b, cp, h, w, dtype, image.device)
    call("miseg_cast_pad", _stream(), _ptr(image), b * h * w, cin, _DT[dtype], _ptr(out), cp)
    return out


def _pack_now(weight: Tensor, dtype, kind: int, ci_begin: int, ci_count: int, packed: Optional[Tensor] = None, cin: Optional[int] = None) -> Tensor:
    cout, cin_w = weight.shape[0], weight.shape[1]
    kind &= 0xff
    if cin is not None and cin != cin_w:      # forward layout of a weight with fewer input channels than the (padded) activation
        assert kind == 0 and cin > cin_w, (kind, cin, cin_w)
        kind, ci_count = cin_w << 8, cin
    else:
        cin = cin_w
    if packed is None:
        packed = torch.empty((ci_count if kind & 0xff else cout) * (cout if kind & 0xff else cin) * 9, dtype=dtype, device=weight.device)
    call("miseg_pack_conv3x3_weights", _stream(), _DT[dtype], _ptr(weight), cout, cin, kind, ci_begin, ci_count, _ptr(packed))
    return packed


_NO_PACK_CACHE = False


class _PackCache:
    """Operand-layout copies of the conv weights, refreshed once per optimiser step in ONE launch.

    The U-Net asks for ~47 packed weight tensors per step (forward layout + one dgrad layout per concat source); they
    only change when the optimiser (or anything else that bumps ``epoch`` / the tensor version) rewrites the masters.
    The first request after such a change re-packs every registered weight with ``miseg_pack_conv3x3_weights_multi``;
    the rest of the step are dictionary hits.  Only parameters that live in a flat buffer are cached (their storage is
    stable); anything else (e.g. the zero-padded stem weight) is packed on the spot."""

    def __init__(self):
        self.epoch = 0            # bumped by FusedAdam.step / FlatBuffers.build / load_state_dict
        self.entries = {}         # key -> entry list, see get()
        self.packed_epoch = -1
        self.jobs_dev = {}        # dtype -> (device job table, njobs, total_blocks, keys)
        self.dirty = True
        self.generation = 0       # bumped whenever the set of packed tensors (hence the addresses a launch tape recorded) changes

    def invalidate(self) -> None:
        self.epoch += 1

    def get(self, weight: Tensor, dtype, kind: int, cb: int, cs: int, cin: Optional[int] = None) -> Tensor:
        """``cin`` (forward layout only): input channels of the PACKED tensor when the weight has fewer (the stem); the kind word then
        carries the weight's own count in its high bits (miseg_pack_conv3x3_weights)."""
        if getattr(weight, "_miseg_grad_slot", None) is None or not weight.is_cuda or _NO_PACK_CACHE:
            return _pack_now(weight, dtype, kind, cb, cs, cin=cin)
        key = (weight.data_ptr(), tuple(weight.shape), dtype, kind, cb, cs)
        ent = self.entries.get(key)
        if ent is None or ent[0]() is not weight:
            # entry: [weakref(weight), dtype, kind, cb, cs, packed, version packed at, epoch packed at]
            ent = self.entries[key] = [weakref.ref(weight), dtype, kind, cb, cs, _pack_now(weight, dtype, kind, cb, cs, cin=cin),
                                       weight._version, self.epoch]
            self.dirty = True
            self.generation += 1
            return ent[5]
        if self.packed_epoch != self.epoch:
            self._repack_all()
        if ent[6] != weight._version or ent[7] != self.epoch:   # edited in place since / registered after the batch
            _pack_now(weight, dtype, kind, cb, cs, ent[5], cin=cin)
            ent[6], ent[7] = weight._version, self.epoch
        return ent[5]

    def _repack_all(self) -> None:
        dead = [k for k, ent in self.entries.items() if ent[0]() is None]   # parameters that no longer exist
        for k in dead:
            del self.entries[k]
        if (dead or self.dirty) and torch.cuda.is_current_stream_capturing():
            # the job table would need a host->device copy, which a capturing stream does not allow: pack one by one
            for ent in self.entries.values():
                w = ent[0]()
                _pack_now(w, ent[1], ent[2] & 0xff, ent[3], ent[4], ent[5], cin=(ent[4] if ent[2] >> 8 else None))
                ent[6], ent[7] = w._version, self.epoch
            self.packed_epoch = self.epoch
            return
        if dead or self.dirty:
            self.jobs_dev = {}
            by_dtype = {}
            for ent in self.entries.values():
                by_dtype.setdefault(ent[1], []).append(ent)
            for dtype, ents in by_dtype.items():
                blob, first = bytearray(), 0
                for ent in ents:
                    w, packed = ent[0](), ent[5]
                    cin_packed = ent[4] if ent[2] >> 8 else w.shape[1]      # stem: packed with more input channels than the weight has
                    blob += struct.pack("<QQiiiiii", w.data_ptr(), packed.data_ptr(), w.shape[0], cin_packed, ent[2], ent[3], ent[4], first)
                    first += (packed.numel() + 255) // 256
                table = torch.frombuffer(blob, dtype=torch.uint8).clone().to(ents[0][5].device)
                self.jobs_dev[dtype] = (table, len(ents), first, ents)
            self.dirty = False
        for dtype, (table, n, blocks, ents) in self.jobs_dev.items():
            live = [ent[0]() for ent in ents]          # strong references for the duration of the launch
            call("miseg_pack_conv3x3_weights_multi", _stream(), _DT[dtype], _ptr(table), n, blocks)
            for ent, w in zip(ents, live):
                ent[6], ent[7] = w._version, self.epoch
        self.packed_epoch = self.epoch


PACK_CACHE = _PackCache()


def _pack(weight: Tensor, dtype, kind: int, ci_begin: int = 0, ci_count: int = 0, cin: Optional[int] = None) -> Tensor:
    if not kind:
        ci_begin, ci_count = 0, weight.shape[1]
        if cin is not None and cin != weight.shape[1]:
            return PACK_CACHE.get(weight, dtype, weight.shape[1] << 8, 0, int(cin), cin=int(cin))
    return PACK_CACHE.get(weight, dtype, kind, int(ci_begin), int(ci_count))


class _ConvBNReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0: Tensor, x1: Optional[Tensor], weight: Tensor, gamma: Tensor, beta: Tensor, running_mean: Tensor,
                running_var: Tensor, nbt: Tensor, training: bool, ups0: int, ups1: int, want_pool: bool, rec: Optional[_BnRec] = None,
                rec0: Optional[_BnRec] = None, rec1: Optional[_BnRec] = None, sink: Optional[_GradSink] = None):
        _need_gpu(x0, x1, weight)
        x0 = as_nhwc(x0)
        dtype, dev = x0.dtype, x0.device
        n, c0 = x0.shape[0], x0.shape[1]
        h, w = x0.shape[2] << ups0, x0.shape[3] << ups0
        c1 = 0
        if x1 is not None:
            x1 = as_nhwc(x1)
            c1 = x1.shape[1]
            assert x1.dtype == dtype and (x1.shape[2] << ups1, x1.shape[3] << ups1) == (h, w)
        cout = weight.shape[0]
        # the stem: one image channel read as a whole (zero-padded) channel vector; the weight keeps its own shape, the pack kernel pads
        assert weight.shape[1] == c0 + c1 or (x1 is None and weight.shape[1] < c0), (weight.shape, c0, c1)
        weight = weight.contiguous().float()
        # the stem (one image channel, padded to a channel vector): its own kernels, no matrix cores (miseg_conv3x3_stem_fwd / _wgrad)
        image = getattr(x0, "_miseg_stem_f32", None)       # stem_input's descriptor: x0 itself is uninitialised
        stem = bool(_STEM_KERNELS and x1 is None and not ups0 and weight.shape[1] == 1 and c0 > 1 and w <= _STEM_MAX_W and
                    query("miseg_conv3x3_stem_supported", _DT[dtype], 1, c0, cout))
        packed = None if stem else _pack(weight, dtype, 0, cin=c0 + c1)
        raw = empty_nhwc(n, cout, h, w, dtype, dev)
        saved = torch.empty(4 * cout, dtype=torch.float32, device=dev)
        counter = SYNC_COUNTERS.take(dev) if training and query("miseg_conv3x3_bn_fwd_fusable", _DT[dtype], c0 + c1, n, h, w, cout) else None
        acc = None
        if stem and training and (counter is not None or not _BN_ACC):
            stem, packed = False, _pack(weight, dtype, 0, cin=c0 + c1)       # (the stem kernel hands its statistics to the accumulator only)
        if image is not None and not stem:                                  # the MFMA path after all: materialise the padded operand
            call("miseg_cast_pad", _stream(), _ptr(image), n * h * w, 1, _DT[dtype], _ptr(x0), c0)
            image = None
        if training and counter is None and _BN_ACC and (stem or query("miseg_conv3x3_fwd_acc_supported", _DT[dtype], c0 + c1, n, h, w, cout)):
            # the statistics leave the convolution as fixed-point atomic adds into one [2 C] accumulator that the step block's upload
            # zeroed; the apply kernel turns them into coefficients itself: no partial rows, no finalize launch
            io = stepio.current()
            # (inside an iteration whose block has no room left: the row-per-block path below; stand-alone use of the layer: a zero fill)
            acc = io.acc64(2 * cout + 2) if io is not None else torch.zeros(2 * cout + 2, dtype=torch.int64, device=dev)   # sums + misfit count
        if training and acc is None:       # rows of the statistics matrix: one per block of the kernel that will serve this shape
            parts = query("miseg_conv3x3_stats_parts", _DT[dtype], c0 + c1, n, h, w) if counter is not None else \
                query("miseg_conv3x3_fwd_parts", _DT[dtype], c0 + c1, n, h, w, cout)
            stats = torch.empty(parts * 2 * cout, dtype=torch.float32, device=dev)
        else:
            parts, stats = 0, None
        es = x0.element_size()
        work = (18.0 * (c0 + c1) * cout * n * h * w, float(es) * n * h * w * (c0 / (4 ** ups0) + c1 / (4 ** ups1) + cout))
        if stem:
            call("miseg_conv3x3_stem_fwd", _stream(), _DT[dtype], _ptr(image if image is not None else x0), int(image is not None), 1 if image is not None else c0,
                 n, h, w, _ptr(weight), 1, cout, _ptr(raw), _ptr(acc),
                 work=(18.0 * cout * n * h * w, float(es) * n * h * w * (c0 + cout)), tag=f"conv3x3_fwd[{h}x{w},{c0 + c1}->{cout}]")
        elif acc is not None:
            call("miseg_conv3x3_fwd_acc", _stream(), _DT[dtype], _ptr(x0), c0, ups0, _ptr(x1), c1, ups1, n, h, w, _ptr(packed), cout, _ptr(raw),
                 _ptr(acc), work=work, tag=f"conv3x3_fwd[{h}x{w},{c0 + c1}->{cout}]")
        elif counter is not None:     # the conv's last block turns the partial sums into `saved` / the running statistics itself
            call("miseg_conv3x3_bn_fwd", _stream(), _DT[dtype], _ptr(x0), c0, ups0, _ptr(x1), c1, ups1, n, h, w, _ptr(packed), cout, _ptr(raw),
                 _ptr(stats), _ptr(gamma), _ptr(beta), BN_EPS, BN_MOMENTUM, _ptr(running_mean), _ptr(running_var), _ptr(nbt), _ptr(saved),
                 _ptr(counter), work=work, tag=f"conv3x3_fwd[{h}x{w},{c0 + c1}->{cout}]")
        else:
            call("miseg_conv3x3_fwd", _stream(), _DT[dtype], _ptr(x0), c0, ups0, _ptr(x1), c1, ups1, n, h, w, _ptr(packed), cout, _ptr(raw),
                 _ptr(stats), work=work, tag=f"conv3x3_fwd[{h}x{w},{c0 + c1}->{cout}]")
        if training and counter is None and acc is None:
            call("miseg_bn_finalize", _stream(), _ptr(stats), parts, cout, n * h * w, _ptr(gamma), _ptr(beta), BN_EPS, BN_MOMENTUM,
                 _ptr(running_mean), _ptr(running_var), _ptr(nbt), _ptr(saved))
        elif not training:
            call("miseg_bn_eval_coeffs", _stream(), cout, _ptr(gamma), _ptr(beta), BN_EPS, _ptr(running_mean), _ptr(running_var), _ptr(saved))
        y = empty_nhwc(n, cout, h, w, dtype, dev)
        pooled = empty_nhwc(n, cout, h // 2, w // 2, dtype, dev) if want_pool else None
        if acc is not None:
            call("miseg_bn_relu_fwd_acc", _stream(), _DT[dtype], _ptr(raw), n, h, w, cout, _ptr(acc), _ptr(gamma), _ptr(beta), BN_EPS, BN_MOMENTUM,
                 _ptr(running_mean), _ptr(running_var), _ptr(nbt), _ptr(saved), _ptr(y), _ptr(pooled),
                 work=(0.0, float(es) * n * h * w * cout * (2.25 if want_pool else 2.0)), tag=f"bn_relu_fwd[{h}x{w},{cout}]")
        else:
            call("miseg_bn_relu_fwd", _stream(), _DT[dtype], _ptr(raw), n, h, w, cout, _ptr(saved), _ptr(y), _ptr(pooled),
                 work=(0.0, float(es) * n * h * w * cout * (2.25 if want_pool else 2.0)), tag=f"bn_relu_fwd[{h}x{w},{cout}]")
        ctx.save_for_backward(x0, x1, weight, gamma, raw, y, saved)
        ctx.param_refs = (weight, gamma, beta)   # the Parameter objects (flat-gradient slots hang off them)
        ctx.cfg = (training, ups0, ups1, want_pool, c0, c1, n, h, w, cout)
        ctx.stem, ctx.image = stem, (image if stem else None)
        if rec is not None:
            rec.raw, rec.saved, rec.shape = raw, saved, (n, cout, h, w)
        ctx.recs = (rec, rec0, rec1)
        ctx.sink = sink
        if want_pool:
            return y, pooled
        return y, None

    @staticmethod
    def backward(ctx, gy: Optional[Tensor], gpool: Optional[Tensor]):
        x0, x1, weight, gamma, raw, y, saved = ctx.saved_tensors
        training, ups0, ups1, want_pool, c0, c1, n, h, w, cout = ctx.cfg
        dtype, dev = raw.dtype, raw.device
        extras = ctx.sink.take() if ctx.sink is not None else []
        if gy is None and gpool is None and not extras:
            return (None,) * 16
        cur = torch.cuda.current_stream(dev)
        for g2, _, _, st2 in extras:
            wait_stream(cur, st2)             # (autograd would have made the same wait before adding the two gradients)
            keep(g2, cur)
        extra = None
        if len(extras) == 1 and (gy is not None or gpool is not None) and extras[0][0].dtype == dtype:
            extra = extras[0]
        elif extras:                          # several heads on one tap, or a tap nothing else consumes: add them here, in torch
            gy = gy.clone() if gy is not None else torch.zeros((n, cout, h, w), dtype=dtype, device=dev, memory_format=torch.channels_last)
            for g2, a, b, _ in extras:
                gy[a:b] += g2.to(dtype)
        rec, rec0, rec1 = ctx.recs
        ext_parts, ext_nparts = rec.take(gy) if rec is not None else (None, 0)
        if _FUSE_BN_BWD and not want_pool and gpool is None and gy is not None and _dgrads_fusable(ctx, dtype):
            return _ConvBNReLU._backward_fused(ctx, gy, ext_parts, ext_nparts)
        gy = None if gy is None else as_nhwc(gy.to(dtype))
        gpool = None if gpool is None else as_nhwc(gpool.to(dtype))
        graw = empty_nhwc(n, cout, h, w, dtype, dev)
        pw, pg, pb = ctx.param_refs
        ggamma, gbeta = grad_slot(pg), grad_slot(pb)   # written in place when the flat gradient buffer has an open slot
        if ggamma is None:
            ggamma = torch.empty(cout, dtype=torch.float32, device=dev)
        if gbeta is None:
            gbeta = torch.empty(cout, dtype=torch.float32, device=dev)
        ws = _ws(query("miseg_bn_bwd_ws_bytes", n, h, w, cout), dev)
        bacc = None
        if _BN_ACC_BWD and not SYNC_COUNTERS.enabled and query("miseg_bn_relu_bwd_acc_supported", _DT[dtype], n, h, w, cout):
            # the reduce kernel adds its sums into a zeroed two-tier fixed-point accumulator, the apply kernel finishes them: no finalize launch
            io = stepio.current()
            bacc = io.acc64(4 * cout + 2) if io is not None else torch.zeros(4 * cout + 2, dtype=torch.int64, device=dev)   # (None: block full -> finalize path)
        if ext_parts is not None and not extras and gpool is None and gy is not None and not want_pool and dtype != torch.float16:
            # the data-gradient kernel that wrote gy has summed dz and dz * xhat per block in its epilogue: finalize + apply only
            call("miseg_bn_relu_bwd_ext", _stream(), _DT[dtype], _ptr(raw), _ptr(gy), n, h, w, cout, _ptr(gamma), _ptr(saved), int(training),
                 _ptr(graw), _ptr(ggamma), _ptr(gbeta), _ptr(ext_parts), ext_nparts, _ptr(ws), ws.numel(),
                 work=(0.0, float(raw.element_size()) * n * h * w * cout * 3.0), tag=f"bn_relu_bwd[{h}x{w},{cout}]")
        elif bacc is not None:
            g2, n0, n1 = (as_nhwc(extra[0]), extra[1], extra[2]) if extra is not None else (None, 0, 0)
            call("miseg_bn_relu_bwd_dual_acc", _stream(), _DT[dtype], _ptr(raw), _ptr(gy), _ptr(gpool), _ptr(g2), n0, n1, n, h, w, cout, _ptr(gamma),
                 _ptr(saved), int(training), _ptr(graw), _ptr(ggamma), _ptr(gbeta), _ptr(ws), ws.numel(), _ptr(bacc),
                 work=(0.0, float(raw.element_size()) * n * h * w * cout * 5.0), tag=f"bn_relu_bwd[{h}x{w},{cout}]")
        elif extra is not None:
            g2, n0, n1, _ = extra
            g2 = as_nhwc(g2)
            call("miseg_bn_relu_bwd_dual", _stream(), _DT[dtype], _ptr(raw), _ptr(gy), _ptr(gpool), _ptr(g2), n0, n1, n, h, w, cout, _ptr(gamma),
                 _ptr(saved), int(training), _ptr(graw), _ptr(ggamma), _ptr(gbeta), _ptr(ws), ws.numel(),
                 work=(0.0, float(raw.element_size()) * n * h * w * cout * 5.0), tag=f"bn_relu_bwd[{h}x{w},{cout}]")
        else:
            call("miseg_bn_relu_bwd_sync", _stream(), _DT[dtype], _ptr(raw), _ptr(y), _ptr(gy), _ptr(gpool), n, h, w, cout, _ptr(gamma), _ptr(saved),
                 int(training), _ptr(graw), _ptr(ggamma), _ptr(gbeta), _ptr(ws), ws.numel(), _ptr(SYNC_COUNTERS.take(dev)),
                 work=(0.0, float(raw.element_size()) * n * h * w * cout * 5.0), tag=f"bn_relu_bwd[{h}x{w},{cout}]")
        gw = None
        if ctx.needs_input_grad[2]:
            gw = grad_slot(pw)
            side = None
            if gw is None:
                gw = torch.empty_like(weight)
            elif _WGRAD_SIDE and (_GRAPH_STREAMS or not torch.cuda.is_current_stream_capturing()):
                side = wgrad_stream(dev)      # only when the result lands in the flat buffer: nothing on this stream touches it again
            cur = torch.cuda.current_stream(dev)
            if side is not None:
                if not _wgrad_dirty:          # first side-stream wgrad of this backward pass: join when the pass ends, so that
                    # param.grad (which already aliases the flat slot) is safe to touch from the main stream as soon as
                    # backward() returns -- gradient accumulation, clip_grad_norm_, inspection -- not only after collect()
                    torch.autograd.Variable._execution_engine.queue_callback(join_wgrad_streams)
                wait_stream(side, cur)        # graw is produced above
                for t in (graw, x0, x1, getattr(ctx, "image", None)):      # keep their memory from being recycled under the side stream
                    keep(t, side)
                _wgrad_dirty.add(dev)
            # launched on the side stream by HANDLE (entering the `torch.cuda.stream` context costs 15-20 us of host time, 22 times per
            # backward pass); the workspace comes from the current stream's pool and is handed to the side stream like the operands
            stem_wg = bool(ctx.stem and not _FUSE_BN_BWD)
            ws2 = _ws(query("miseg_conv3x3_stem_wgrad_ws_bytes", cout) if stem_wg else query("miseg_conv3x3_wgrad_ws_bytes", n, h, w, c0 + c1, cout), dev)
            # the stem's weight has fewer input channels than the channel vector its activation was padded to: the kernel computes the
            # padded gradient, a slice of it is the parameter's
            gw_full = gw if (weight.shape[1] == c0 + c1 or stem_wg) else torch.empty((cout, c0 + c1, 3, 3), dtype=torch.float32, device=dev)
            if side is not None:
                keep(ws2, side)
                if gw_full is not gw:
                    keep(gw_full, side)

            def launch(stream_handle):
                if stem_wg:      # the last kernel of the backward pass: a pixel-per-thread kernel, gw [cout][1][3][3] written directly
                    img = ctx.image
                    call("miseg_conv3x3_stem_wgrad", stream_handle, _DT[dtype], _ptr(img if img is not None else x0), int(img is not None),
                         1 if img is not None else c0, n, h, w, _ptr(graw), cout, _ptr(gw), _ptr(ws2), ws2.numel(),
                         work=(18.0 * cout * n * h * w, float(raw.element_size()) * n * h * w * (c0 + cout)), tag=f"conv3x3_wgrad[{h}x{w},{c0 + c1}->{cout}]")
                    return
                call("miseg_conv3x3_wgrad", stream_handle, _DT[dtype], _ptr(x0), c0, ups0, _ptr(x1), c1, ups1, n, h, w, _ptr(graw), cout, _ptr(gw_full),
                     _ptr(ws2), ws2.numel(), work=(18.0 * (c0 + c1) * cout * n * h * w, float(raw.element_size()) * n * h * w * (c0 + c1 + cout)),
                     tag=f"conv3x3_wgrad[{h}x{w},{c0 + c1}->{cout}]")
                if gw_full is not gw:
                    call("miseg_conv3x3_wgrad_slice", stream_handle, _ptr(gw_full), cout, c0 + c1, weight.shape[1], _ptr(gw))
            from . import _cabi
            if side is None:
                launch(_stream())
            elif _cabi.TIMER is not None:       # the per-kernel timer records its events on torch's current stream
                with torch.cuda.stream(side):
                    launch(_stream())
            else:
                launch(side.cuda_stream)
        grads = [None, None]
        from .ops import _GradJoin
        if _DUAL_DGRAD and x1 is not None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and not ups0 and not ups1 and \
                c0 % 16 == 0 and c1 % 4 == 0 and not _GradJoin.enabled and not (_FUSE_BN_BWD and _FUSE_BN_RED) and not _RED_ONLY:
            # concat of two full-resolution sources: both data gradients in ONE launch (graw read once, twice the blocks)
            packed = _pack(weight, dtype, 1, 0, c0 + c1)
            g0, g1 = empty_nhwc(n, c0, h, w, dtype, dev), empty_nhwc(n, c1, h, w, dtype, dev)
            call("miseg_conv3x3_dgrad_dual", _stream(), _DT[dtype], _ptr(graw), cout, n, h, w, _ptr(packed), c0, _ptr(g0), c1, _ptr(g1),
                 work=(18.0 * (c0 + c1) * cout * n * h * w, float(raw.element_size()) * n * h * w * (c0 + c1 + cout)),
                 tag=f"conv3x3_dgrad[{h}x{w},{cout}->{c0}+{c1}]")
            return g0, g1, gw, ggamma, gbeta, None, None, None, None, None, None, None, None, None, None, None
        for s, (cb, cs, ups, xs) in enumerate(((0, c0, ups0, x0), (c0, c1, ups1, x1))):
            if xs is None or not ctx.needs_input_grad[s]:
                continue
            packed = _pack(weight, dtype, 1, cb, cs)
            if ups and _FUSE_UPS_DGRAD and query("miseg_conv3x3_fwd_sumpool_supported", _DT[dtype], cout, n, h, w, cs):
                # the source was read through the x2 upsample: its gradient is the 2x2 sum-pool of the data gradient -- pooled in the
                # convolution's epilogue, the full-resolution gradient (4x the bytes) never exists
                from .ops import _GradJoin
                joined = _GradJoin.take(xs) if query("miseg_conv3x3_fwd_sumpool_acc_supported", _DT[dtype], cout, n, h, w) else None
                if joined is not None:
                    # the source is also a local-MI tap and its head's backward has already written its gradient: add ours into it
                    gsum, written, writer = joined
                    cur_s = torch.cuda.current_stream(dev)
                    cur_s.wait_event(written)
                    gsum.record_stream(cur_s)
                    call("miseg_conv3x3_fwd_sumpool_acc", _stream(), _DT[dtype], _ptr(graw), cout, n, h, w, _ptr(packed), cs, _ptr(gsum),
                         work=(18.0 * cs * cout * n * h * w, float(raw.element_size()) * n * h * w * (cs / 2 + cout)),
                         tag=f"conv3x3_dgrad[{h}x{w},{cout}->{cs}]")
                    if writer != cur_s:
                        done = torch.cuda.Event()
                        done.record(cur_s)
                        writer.wait_event(done)
                    grads[s] = None
                    continue
                glow = empty_nhwc(n, cs, h // 2, w // 2, dtype, dev)
                call("miseg_conv3x3_fwd_sumpool", _stream(), _DT[dtype], _ptr(graw), cout, n, h, w, _ptr(packed), cs, _ptr(glow),
                     work=(18.0 * cs * cout * n * h * w, float(raw.element_size()) * n * h * w * (cs / 4 + cout)),
                     tag=f"conv3x3_dgrad[{h}x{w},{cout}->{cs}]")
                _GradJoin.offer(xs, glow)      # if the source is a tap whose head backward has yet to run, it adds into this tensor
                grads[s] = glow
                continue
            gfull = empty_nhwc(n, cs, h, w, dtype, dev)
            src_rec = (rec0, rec1)[s]
            red = _red_target(src_rec, ups, dtype, cout, n, h, w, cs)
            if red is not None:      # the source is a BatchNorm layer read at full resolution: its backward sums in this launch's epilogue
                parts, nparts = red
                call("miseg_conv3x3_dgrad_bn", _stream(), _DT[dtype], _ptr(graw), None, None, cout, n, h, w, _ptr(packed), cs, _ptr(gfull), 0,
                     _ptr(src_rec.raw), _ptr(src_rec.saved), _ptr(parts),
                     work=(18.0 * cs * cout * n * h * w, float(raw.element_size()) * n * h * w * (2 * cs + cout)), tag=f"conv3x3_dgrad[{h}x{w},{cout}->{cs}]")
                src_rec.offer(gfull, parts, nparts)
                grads[s] = gfull
                continue
            call("miseg_conv3x3_fwd", _stream(), _DT[dtype], _ptr(graw), cout, 0, None, 0, 0, n, h, w, _ptr(packed), cs, _ptr(gfull), None,
                 work=(18.0 * cs * cout * n * h * w, float(raw.element_size()) * n * h * w * (cs + cout)), tag=f"conv3x3_dgrad[{h}x{w},{cout}->{cs}]")
            if ups:
                glow = empty_nhwc(n, cs, h // 2, w // 2, dtype, dev)
                call("miseg_sumpool2x2", _stream(), _DT[dtype], _ptr(gfull), n, h, w, cs, _ptr(glow), 0)
                gfull = glow
            grads[s] = gfull
        return grads[0], grads[1], gw, ggamma, gbeta, None, None, None, None, None, None, None, None, None, None, None

    @staticmethod
    def _backward_fused(ctx, gy_in: Tensor, ext_parts: Optional[Tensor], ext_nparts: int):
        x0, x1, weight, gamma, raw, y, saved = ctx.saved_tensors
        training, ups0, ups1, want_pool, c0, c1, n, h, w, cout = ctx.cfg
        rec, rec0, rec1 = ctx.recs
        dtype, dev, es = raw.dtype, raw.device, raw.element_size()
        gy = as_nhwc(gy_in.to(dtype))
        pw, pg, pb = ctx.param_refs
        ggamma, gbeta = grad_slot(pg), grad_slot(pb)
        if ggamma is None:
            ggamma = torch.empty(cout, dtype=torch.float32, device=dev)
        if gbeta is None:
            gbeta = torch.empty(cout, dtype=torch.float32, device=dev)
        coef = torch.empty(6 * cout, dtype=torch.float32, device=dev)
        if ext_parts is not None:       # the consumer's data-gradient kernel already summed dz and dz * xhat per block: finalize only
            call("miseg_bn_relu_bwd_stats", _stream(), _DT[dtype], None, None, n, h, w, cout, _ptr(gamma), _ptr(saved), int(training), _ptr(coef),
                 _ptr(ggamma), _ptr(gbeta), _ptr(ext_parts), ext_nparts, None, 0)
        else:
            ws = _ws(query("miseg_bn_bwd_ws_bytes", n, h, w, cout), dev)
            call("miseg_bn_relu_bwd_stats", _stream(), _DT[dtype], _ptr(raw), _ptr(gy), n, h, w, cout, _ptr(gamma), _ptr(saved), int(training),
                 _ptr(coef), _ptr(ggamma), _ptr(gbeta), None, 0, _ptr(ws), ws.numel(),
                 work=(0.0, float(es) * n * h * w * cout * 2.0), tag=f"bn_relu_bwd[{h}x{w},{cout}]")
        gw = None
        if ctx.needs_input_grad[2]:
            gw = grad_slot(pw)
            side = None
            if gw is None:
                gw = torch.empty_like(weight)
            elif _WGRAD_SIDE and (_GRAPH_STREAMS or not torch.cuda.is_current_stream_capturing()):
                side = wgrad_stream(dev)
            cur = torch.cuda.current_stream(dev)
            if side is not None:
                if not _wgrad_dirty:
                    torch.autograd.Variable._execution_engine.queue_callback(join_wgrad_streams)
                wait_stream(side, cur)        # coef is produced above
                for t in (raw, gy, coef, x0, x1):
                    keep(t, side)
                _wgrad_dirty.add(dev)
            with torch.cuda.stream(side if side is not None else cur):
                ws2 = _ws(query("miseg_conv3x3_wgrad_ws_bytes", n, h, w, c0 + c1, cout), dev)
                call("miseg_conv3x3_wgrad_bn", _stream(), _DT[dtype], _ptr(x0), c0, ups0, _ptr(x1), c1, ups1, n, h, w, _ptr(raw), _ptr(gy), _ptr(coef),
                     cout, _ptr(gw), _ptr(ws2), ws2.numel(),
                     work=(18.0 * (c0 + c1) * cout * n * h * w, float(es) * n * h * w * (c0 + c1 + 2 * cout)), tag=f"conv3x3_wgrad[{h}x{w},{c0 + c1}->{cout}]")
        grads = [None, None]
        for s, (cb, cs, ups, xs, src_rec) in enumerate(((0, c0, ups0, x0, rec0), (c0, c1, ups1, x1, rec1))):
            if xs is None or not ctx.needs_input_grad[s]:
                continue
            packed = _pack(weight, dtype, 1, cb, cs)
            flops = 18.0 * cs * cout * n * h * w
            pooled = bool(ups) and bool(query("miseg_conv3x3_dgrad_bn_supported", _DT[dtype], cout, n, h, w, cs, 1))
            if pooled:
                glow = empty_nhwc(n, cs, h // 2, w // 2, dtype, dev)
                call("miseg_conv3x3_dgrad_bn", _stream(), _DT[dtype], _ptr(raw), _ptr(gy), _ptr(coef), cout, n, h, w, _ptr(packed), cs, _ptr(glow), 1,
                     None, None, None, work=(flops, float(es) * n * h * w * (cs / 4 + 2 * cout)), tag=f"conv3x3_dgrad[{h}x{w},{cout}->{cs}]")
                grads[s] = glow
                continue
            gfull = empty_nhwc(n, cs, h, w, dtype, dev)
            red = _red_target(src_rec, ups, dtype, cout, n, h, w, cs)
            if red is not None:
                parts, nparts = red
                call("miseg_conv3x3_dgrad_bn", _stream(), _DT[dtype], _ptr(raw), _ptr(gy), _ptr(coef), cout, n, h, w, _ptr(packed), cs, _ptr(gfull), 0,
                     _ptr(src_rec.raw), _ptr(src_rec.saved), _ptr(parts), work=(flops, float(es) * n * h * w * (2 * cs + 2 * cout)),
                     tag=f"conv3x3_dgrad[{h}x{w},{cout}->{cs}]")
                src_rec.offer(gfull, parts, nparts)
            else:
                call("miseg_conv3x3_dgrad_bn", _stream(), _DT[dtype], _ptr(raw), _ptr(gy), _ptr(coef), cout, n, h, w, _ptr(packed), cs, _ptr(gfull), 0,
                     None, None, None, work=(flops, float(es) * n * h * w * (cs + 2 * cout)), tag=f"conv3x3_dgrad[{h}x{w},{cout}->{cs}]")
            if ups:
                glow = empty_nhwc(n, cs, h // 2, w // 2, dtype, dev)
                call("miseg_sumpool2x2", _stream(), _DT[dtype], _ptr(gfull), n, h, w, cs, _ptr(glow), 0)
                gfull = glow
            grads[s] = gfull
        return grads[0], grads[1], gw, ggamma, gbeta, None, None, None, None, None, None, None, None, None, None, None


def _dgrads_fusable(ctx, dtype) -> bool:
    """Can every data-gradient launch of this layer's backward form graw in its loader?"""
    from .ops import _GradJoin
    training, ups0, ups1, want_pool, c0, c1, n, h, w, cout = ctx.cfg
    if _GradJoin.enabled or cout % vec_of(dtype):
        return False
    return all(cs == 0 or query("miseg_conv3x3_dgrad_bn_supported", _DT[dtype], cout, n, h, w, cs, 0) for cs in (c0, c1))


def _red_target(src_rec: Optional[_BnRec], ups: int, dtype, cout: int, n: int, h: int, w: int, cs: int):
    """(parts, nparts) if this data-gradient launch can take the source layer's BatchNorm-backward sums in its epilogue."""
    if not ((_FUSE_BN_BWD and _FUSE_BN_RED) or _RED_ONLY) or src_rec is None or ups or src_rec.raw is None or src_rec.shape != (n, cs, h, w) or \
            src_rec.raw.dtype != dtype:
        return None
    nparts = query("miseg_conv3x3_dgrad_red_parts", _DT[dtype], cout, n, h, w, cs)
    if not nparts:
        return None
    return torch.empty(nparts * 2 * cs, dtype=torch.float32, device=src_rec.raw.device), nparts


def conv_bn_relu(x0: Tensor, x1: Optional[Tensor], weight: Tensor, gamma: Tensor, beta: Tensor, running_mean: Tensor,
                 running_var: Tensor, nbt: Tensor, training: bool, ups0: int = 0, ups1: int = 0,
                 want_pool: bool = False) -> Tuple[Tensor, Optional[Tensor]]:
    # the hand-over record exists only where it can be used: the opt-in fused backward, a layer without the fused pool (a pooled layer's
    # backward routes through the 2x2 windows and keeps its own reduce), a forward pass that records a graph
    rec = _BnRec() if (((_FUSE_BN_BWD and _FUSE_BN_RED) or _RED_ONLY) and not want_pool and torch.is_grad_enabled()) else None
    rec0 = getattr(x0, "_miseg_bn", None) if not ups0 else None
    rec1 = getattr(x1, "_miseg_bn", None) if x1 is not None and not ups1 else None
    sink = _GradSink() if torch.is_grad_enabled() and not _FUSE_BN_BWD else None
    y, pooled = _ConvBNReLU.apply(x0, x1, weight, gamma, beta, running_mean, running_var, nbt, bool(training), int(ups0), int(ups1),
                                  bool(want_pool), rec, rec0, rec1, sink)
    if rec is not None and rec.raw is not None:
        y._miseg_bn = rec
    if sink is not None and y.requires_grad:
        y._miseg_grad_sink = sink
    return y, pooled


class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor):
        _need_gpu(x, weight, bias)
        x = as_nhwc(x)
        n, cin, h, w = x.shape
        cout = weight.shape[0]
        wf = weight.contiguous().float().view(cout, cin)
        bf = bias.contiguous().float()
        out = empty_nhwc(n, cout, h, w, torch.float32, x.device)
        call("miseg_conv1x1_fwd", _stream(), _DT[x.dtype], _ptr(x), n, h, w, cin, _ptr(wf), _ptr(bf), cout, _ptr(out))
        from .ops import _GradJoin
        _GradJoin.clear()                # offers of an earlier backward pass that nobody took
        ctx.save_for_backward(x, wf)
        ctx.wshape = tuple(weight.shape)
        ctx.param_refs = (weight, bias)
        return out

    @staticmethod
    def backward(ctx, gout: Tensor):
        x, wf = ctx.saved_tensors
        n, cin, h, w = x.shape
        cout = wf.shape[0]
        gout = as_nhwc(gout.float())
        gin = empty_nhwc(n, cin, h, w, x.dtype, x.device) if ctx.needs_input_grad[0] else None
        pw, pb = ctx.param_refs
        gw, gb = grad_slot(pw), grad_slot(pb)       # written in place when the flat gradient buffer has an open slot
        gw = gw.view(cout, cin) if gw is not None else torch.empty_like(wf)
        if gb is None:
            gb = torch.empty(cout, dtype=torch.float32, device=x.device)
        ws = _ws(query("miseg_conv1x1_bwd_ws_bytes", n, h, w, cin, cout), x.device)
        call("miseg_conv1x1_bwd", _stream(), _DT[x.dtype], _ptr(x), _ptr(gout), n, h, w, cin, _ptr(wf), cout, _ptr(gin), _ptr(gw), _ptr(gb),
             _ptr(ws), ws.numel())
        if gin is not None:
            from .ops import _GradJoin
            _GradJoin.offer(x, gin)      # a tap's head backward may add its gradient of x into this tensor (ops._GradJoin)
        return gin, gw.view(ctx.wshape), gb


def conv1x1(x: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    return _Conv1x1.apply(x, weight, bias)


def count_nonfinite(grad: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """Device float[1]: how many entries of the fp32 tensor ``grad`` are inf or NaN (``miseg_count_nonfinite``)."""
    _need_gpu(grad)
    assert grad.dtype == torch.float32 and grad.is_contiguous()
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=grad.device)
    call("miseg_count_nonfinite", _stream(), _ptr(grad), grad.numel(), _ptr(out))
    return out


def adam_step(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, hyper: Tensor, beta1: float, beta2: float,
              grad_scale: float = 1.0, guard: Optional[Tensor] = None) -> None:
    """Fused Adam on flat fp32 buffers; ``hyper`` = device fp32[4] (lr/bc1, 1/sqrt(bc2), eps, weight_decay); ``grad`` is read as
    grad / grad_scale (the static loss scale of the fp16 mode); ``guard`` = fp32 device flags, any non-zero / NaN one turns the launch
    into a no-op (a failed deferred check must not move the weights)."""
    _need_gpu(param, grad, exp_avg, exp_avg_sq, hyper)
    if guard is not None:
        _need_gpu(guard)
        assert guard.dtype == torch.float32 and guard.is_contiguous()
    # grad_scale == -1: the kernel reads 1 / loss scale from hyper[4] (the step block: a dynamic scale under a replayed launch tape)
    call("miseg_adam_step_guarded", _stream(), _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), float(beta1),
         float(beta2), _ptr(hyper), float(grad_scale), _ptr(guard), 0 if guard is None else guard.numel())
