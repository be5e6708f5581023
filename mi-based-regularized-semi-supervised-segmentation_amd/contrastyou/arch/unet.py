"""2D U-Net on the MI355X kernels, drop-in for the reference's ``contrastyou.arch.UNet``.

Same constructor, attributes, sub-module names (forward hooks work on any name in
``component_names``) and state_dict keys as ref ``contrastyou/arch/unet.py:44-194``, so reference
checkpoints load unchanged.  The ``nn.Conv2d`` / ``nn.BatchNorm2d`` children are parameter holders
only: every block's forward is one fused HIP sequence (conv3x3 -> batch statistics -> BN+ReLU
[+2x2 max-pool]) from ``miseg_amd.unet_ops``; nearest-x2 upsampling and the skip ``torch.cat`` are
index math inside the consumer conv, never materialised.  Activations are channels_last tensors of
``compute_dtype`` (float32 = exact parity mode, bfloat16 = MFMA bf16 operands, fp32 accumulate).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional, Sequence, Tuple, Union

import torch
from torch import Tensor, nn

from miseg_amd import unet_ops

__all__ = ["UNet"]

_DTYPES = {"float32": torch.float32, "fp32": torch.float32, "bfloat16": torch.bfloat16, "bf16": torch.bfloat16,
           "float16": torch.float16, "fp16": torch.float16, "half": torch.float16}


def _as_dtype(d) -> torch.dtype:
    return d if isinstance(d, torch.dtype) else _DTYPES[str(d).lower()]


def _cbr(holder_conv: nn.Conv2d, holder_bn: nn.BatchNorm2d, x0: Tensor, x1: Optional[Tensor], training: bool, vec: int,
         ups0: int = 0, ups1: int = 0, want_pool: bool = False):
    weight = holder_conv.weight     # (the stem's has fewer input channels than the zero-padded activation: the pack kernel pads it)
    return unet_ops.conv_bn_relu(x0, x1, weight, holder_bn.weight, holder_bn.bias, holder_bn.running_mean, holder_bn.running_var,
                                 holder_bn.num_batches_tracked, training, ups0, ups1, want_pool)


class conv_block(nn.Module):
    """(conv3x3 no-bias -> BN -> ReLU) x 2   (ref unet.py:10-25).  Accepts one tensor or a (skip, up) pair."""

    def __init__(self, in_ch: int, out_ch: int, compute_dtype=torch.float32, pool_after: bool = False):
        super().__init__()
        layers = []
        for cin in (in_ch, out_ch):
            layers += [nn.Conv2d(cin, out_ch, kernel_size=3, stride=1, padding=1, bias=False), nn.BatchNorm2d(out_ch),
                       nn.ReLU(inplace=True)]
        self.conv = nn.Sequential(*layers)
        self.compute_dtype = compute_dtype
        self.pool_after = pool_after
        self.stem = in_ch % unet_ops.vec_of(compute_dtype) != 0

    def forward(self, x: Union[Tensor, Tuple[Tensor, Tensor]]) -> Tensor:
        x0, x1 = (x if isinstance(x, (tuple, list)) else (x, None))
        if self.stem:
            x0 = unet_ops.stem_input(x0, self.compute_dtype)
        vec = unet_ops.vec_of(self.compute_dtype)
        h, _ = _cbr(self.conv[0], self.conv[1], x0, x1, self.training, vec)
        y, pooled = _cbr(self.conv[3], self.conv[4], h, None, self.training, vec, want_pool=self.pool_after)
        if pooled is not None:
            y._miseg_pooled = pooled  # picked up by the MaxPool that follows (fused into this block's epilogue)
        return y


class up_conv(nn.Module):
    """nearest x2 upsample -> conv3x3 no-bias -> BN -> ReLU   (ref unet.py:28-40); the upsample is fused."""

    def __init__(self, in_ch: int, out_ch: int, compute_dtype=torch.float32):
        super().__init__()
        self.up = nn.Sequential(nn.Upsample(scale_factor=2), nn.Conv2d(in_ch, out_ch, kernel_size=3, stride=1, padding=1, bias=False),
                                nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True))
        self.compute_dtype = compute_dtype

    def forward(self, x: Tensor) -> Tensor:
        y, _ = _cbr(self.up[1], self.up[2], x, None, self.training, unet_ops.vec_of(self.compute_dtype), ups0=1)
        return y


class _FusedMaxPool(nn.MaxPool2d):
    """MaxPool2d(2,2) whose result was already produced by the preceding block's fused epilogue."""

    def forward(self, x: Tensor) -> Tensor:
        pooled = getattr(x, "_miseg_pooled", None)
        return pooled if pooled is not None else super().forward(x)


class UNet(nn.Module):
    dimension_dict = {"Conv1": 16, "Conv2": 32, "Conv3": 64, "Conv4": 128, "Conv5": 256,
                      "Up_conv5": 128, "Up_conv4": 64, "Up_conv3": 32, "Up_conv2": 16}

    def __init__(self, input_dim: int = 3, num_classes: int = 1, compute_dtype="float32", mi_precision: Optional[str] = None):
        """``compute_dtype`` / ``mi_precision`` are this repo's two additions to ``config/semi.yaml``'s ``Arch`` section (ref
        config/semi.yaml:3-5 has input_dim and num_classes): the storage / MFMA operand type of the network, and the arithmetic of the
        local-MI contraction that the trainers apply through ``miseg_amd.ops.resolve_mi_precision`` (default by compute_dtype)."""
        super().__init__()
        self.input_dim, self.num_classes = input_dim, num_classes
        self.compute_dtype = dt = _as_dtype(compute_dtype)
        self.mi_precision = None if mi_precision in (None, "", "auto") else str(mi_precision)
        widths = [16, 32, 64, 128, 256]
        for i in range(1, 5):
            setattr(self, f"Maxpool{i}", _FusedMaxPool(kernel_size=2, stride=2))
        cin = input_dim
        for i, wdt in enumerate(widths, start=1):
            setattr(self, f"Conv{i}", conv_block(cin, wdt, dt, pool_after=i < 5))
            cin = wdt
        for lvl in (5, 4, 3, 2):
            wdt = widths[lvl - 2]
            setattr(self, f"Up{lvl}", up_conv(2 * wdt, wdt, dt))
            setattr(self, f"Up_conv{lvl}", conv_block(2 * wdt, wdt, dt))
        self.DeConv_1x1 = nn.Conv2d(16, num_classes, kernel_size=1, stride=1, padding=0)

    def forward(self, x: Tensor, return_features: bool = False):
        enc = []
        h = x
        for i in range(1, 6):
            if i > 1:
                h = getattr(self, f"Maxpool{i - 1}")(h)
            h = getattr(self, f"Conv{i}")(h)
            enc.append(h)
        dec = []
        d = enc[4]
        for lvl in (5, 4, 3, 2):
            up = getattr(self, f"Up{lvl}")(d)
            d = getattr(self, f"Up_conv{lvl}")((enc[lvl - 2], up))   # == conv_block(torch.cat((skip, up), 1))
            dec.append(d)
        logits = self._head(d)
        if return_features:
            return logits, tuple(reversed(enc)), tuple(dec)
        return logits

    def _head(self, d: Tensor) -> Tensor:
        # route through the holder's __call__ machinery so hooks registered on DeConv_1x1 fire
        holder = self.DeConv_1x1
        if holder._forward_hooks or holder._forward_pre_hooks:
            return _HookedHead.apply_with_hooks(holder, d)
        return unet_ops.conv1x1(d, holder.weight, holder.bias)

    # ---- gradient gating by component range (ref unet.py:135-182)
    def _set_grad(self, names: Sequence[str], flag: bool) -> None:
        for n in names:
            for p in getattr(self, n).parameters():
                p.requires_grad = flag

    def _range(self, from_: str, util: str):
        names = self.component_names
        assert from_ in names, from_
        assert util in names, util
        lo, hi = names.index(from_), names.index(util)
        assert lo <= hi, (from_, util)
        return names[lo:hi + 1]

    def enable_grad(self, from_: str, util: str):
        self._set_grad(self._range(from_, util), True)

    def disable_grad(self, from_: str, util: str):
        self._set_grad(self._range(from_, util), False)

    def enable_grad_util(self, name: str):
        self._set_grad(self._range(self.component_names[0], name), True)

    def disable_grad_util(self, name: str):
        self._set_grad(self._range(self.component_names[0], name), False)

    def enable_grad_encoder(self):
        self._set_grad(self.encoder_names, True)

    def disable_grad_encoder(self):
        self._set_grad(self.encoder_names, False)

    def enable_grad_decoder(self):
        self._set_grad(self.decoder_names, True)

    def disable_grad_decoder(self):
        self._set_grad(self.decoder_names, False)

    def enable_grad_all(self):
        self._set_grad(self.component_names, True)

    def disable_grad_all(self):
        self._set_grad(self.component_names, False)

    @property
    def encoder_names(self):
        return [f"Conv{i}" for i in range(1, 6)]

    @property
    def decoder_names(self):
        names = []
        for lvl in (5, 4, 3, 2):
            names += [f"Up{lvl}", f"Up_conv{lvl}"]
        return names + ["DeConv_1x1"]

    @property
    def component_names(self):
        return self.encoder_names + self.decoder_names

    def weight_norm(self):
        return OrderedDict((name, p.norm().item()) for name, p in self.named_parameters())


class _HookedHead:
    @staticmethod
    def apply_with_hooks(holder: nn.Conv2d, d: Tensor) -> Tensor:
        orig = holder.forward
        holder.forward = lambda inp: unet_ops.conv1x1(inp, holder.weight, holder.bias)
        try:
            return holder(d)
        finally:
            holder.forward = orig
