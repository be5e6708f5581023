"""Oracle: 2D U-Net forward as pure functions over a reference-compatible state_dict.

Restates ``contrastyou/arch/unet.py``: conv_block :10-25 (conv3x3 no-bias -> BN -> ReLU, twice),
up_conv :28-40 (nearest x2 -> conv3x3 no-bias -> BN -> ReLU), UNet widths :44-54,66-84,
forward wiring :86-133.  Parameters live in a flat ``dict[str, Tensor]`` with the
reference's state_dict keys (``Conv1.conv.0.weight`` ... ``DeConv_1x1.bias``) so a
reference checkpoint drops straight in.  Autograd works through it (used for golden
gradient checks).  Test infrastructure only.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

ENCODER = ["Conv1", "Conv2", "Conv3", "Conv4", "Conv5"]
DECODER = ["Up5", "Up_conv5", "Up4", "Up_conv4", "Up3", "Up_conv3", "Up2", "Up_conv2", "DeConv_1x1"]
WIDTHS = {"Conv1": 16, "Conv2": 32, "Conv3": 64, "Conv4": 128, "Conv5": 256,
          "Up_conv5": 128, "Up_conv4": 64, "Up_conv3": 32, "Up_conv2": 16}
BN_EPS = 1e-5  # nn.BatchNorm2d default
BN_MOMENTUM = 0.1


def block_channels(input_dim: int) -> "OrderedDict[str, tuple[int, int]]":
    """(in_ch, out_ch) of every named sub-module in execution order (unet.py:66-84)."""
    return OrderedDict([
        ("Conv1", (input_dim, 16)), ("Conv2", (16, 32)), ("Conv3", (32, 64)), ("Conv4", (64, 128)),
        ("Conv5", (128, 256)),
        ("Up5", (256, 128)), ("Up_conv5", (256, 128)), ("Up4", (128, 64)), ("Up_conv4", (128, 64)),
        ("Up3", (64, 32)), ("Up_conv3", (64, 32)), ("Up2", (32, 16)), ("Up_conv2", (32, 16)),
    ])


def init_state(input_dim: int = 1, num_classes: int = 4, seed: int = 0, dtype=torch.float32) -> "OrderedDict[str, Tensor]":
    """Deterministic random state with the reference key layout (values are NOT the
    reference's init; goldens carry their own state_dict)."""
    rs = np.random.RandomState(seed)  # legacy MT19937 stream: stable across numpy/torch versions
    sd: "OrderedDict[str, Tensor]" = OrderedDict()

    def randn(*shape):
        return torch.from_numpy(rs.standard_normal(shape))

    def conv(prefix, cin, cout):
        sd[f"{prefix}.weight"] = (randn(cout, cin, 3, 3) * (2.0 / (cin * 9)) ** 0.5).to(dtype)

    def bn(prefix, c):
        sd[f"{prefix}.weight"] = (1.0 + 0.1 * randn(c)).to(dtype)
        sd[f"{prefix}.bias"] = (0.1 * randn(c)).to(dtype)
        sd[f"{prefix}.running_mean"] = torch.zeros(c, dtype=dtype)
        sd[f"{prefix}.running_var"] = torch.ones(c, dtype=dtype)
        sd[f"{prefix}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    for name, (cin, cout) in block_channels(input_dim).items():
        if name.startswith("Up") and not name.startswith("Up_conv"):
            conv(f"{name}.up.1", cin, cout)
            bn(f"{name}.up.2", cout)
        else:
            conv(f"{name}.conv.0", cin, cout)
            bn(f"{name}.conv.1", cout)
            conv(f"{name}.conv.3", cout, cout)
            bn(f"{name}.conv.4", cout)
    sd["DeConv_1x1.weight"] = (randn(num_classes, 16, 1, 1) * 0.25).to(dtype)
    sd["DeConv_1x1.bias"] = (0.1 * randn(num_classes)).to(dtype)
    return sd


def _conv_bn_relu(sd, conv_key: str, bn_key: str, x: Tensor, training: bool, update_stats: bool) -> Tensor:
    """One conv3x3(no bias) -> BatchNorm2d -> ReLU (unet.py:15-17 / :33-35)."""
    y = F.conv2d(x, sd[f"{conv_key}.weight"], None, stride=1, padding=1)
    rm, rv = sd[f"{bn_key}.running_mean"], sd[f"{bn_key}.running_var"]
    if training:
        if update_stats:
            y = F.batch_norm(y, rm, rv, sd[f"{bn_key}.weight"], sd[f"{bn_key}.bias"], True, BN_MOMENTUM, BN_EPS)
            sd[f"{bn_key}.num_batches_tracked"] += 1
        else:
            y = F.batch_norm(y, None, None, sd[f"{bn_key}.weight"], sd[f"{bn_key}.bias"], True, BN_MOMENTUM, BN_EPS)
    else:
        y = F.batch_norm(y, rm, rv, sd[f"{bn_key}.weight"], sd[f"{bn_key}.bias"], False, BN_MOMENTUM, BN_EPS)
    return F.relu(y)


def conv_block(sd, name: str, x: Tensor, training: bool, update_stats: bool = True) -> Tensor:
    x = _conv_bn_relu(sd, f"{name}.conv.0", f"{name}.conv.1", x, training, update_stats)
    return _conv_bn_relu(sd, f"{name}.conv.3", f"{name}.conv.4", x, training, update_stats)


def up_conv(sd, name: str, x: Tensor, training: bool, update_stats: bool = True) -> Tensor:
    x = F.interpolate(x, scale_factor=2, mode="nearest")  # nn.Upsample(scale_factor=2), unet.py:32
    return _conv_bn_relu(sd, f"{name}.up.1", f"{name}.up.2", x, training, update_stats)


def unet_forward(sd, x: Tensor, training: bool = True, update_stats: bool = True):
    """UNet.forward (unet.py:86-133).  Returns ``(logits, feats)`` where ``feats`` maps every
    component name (unet.py:184-194) to that sub-module's output -- what a forward hook sees."""
    f: "OrderedDict[str, Tensor]" = OrderedDict()
    pool = lambda t: F.max_pool2d(t, kernel_size=2, stride=2)  # noqa: E731  (unet.py:61-64)
    e1 = f["Conv1"] = conv_block(sd, "Conv1", x, training, update_stats)
    e2 = f["Conv2"] = conv_block(sd, "Conv2", pool(e1), training, update_stats)
    e3 = f["Conv3"] = conv_block(sd, "Conv3", pool(e2), training, update_stats)
    e4 = f["Conv4"] = conv_block(sd, "Conv4", pool(e3), training, update_stats)
    e5 = f["Conv5"] = conv_block(sd, "Conv5", pool(e4), training, update_stats)
    d = e5
    for lvl, skip in ((5, e4), (4, e3), (3, e2), (2, e1)):
        u = f[f"Up{lvl}"] = up_conv(sd, f"Up{lvl}", d, training, update_stats)
        d = f[f"Up_conv{lvl}"] = conv_block(sd, f"Up_conv{lvl}", torch.cat((skip, u), dim=1), training, update_stats)
    logits = f["DeConv_1x1"] = F.conv2d(d, sd["DeConv_1x1.weight"], sd["DeConv_1x1.bias"])
    return logits, f


def trainable_keys(sd) -> list[str]:
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


def unet_forward_bf16_emulated(sd, x: Tensor):
    """Training-mode forward with the bf16 kernels' rounding points reproduced in fp32 arithmetic: operands
    (activations, weights) rounded to bf16, fp32 accumulation, batch statistics from the un-rounded fp32
    accumulators, the raw conv output and every stored activation rounded to bf16, logits fp32.  Used to check
    the bf16 HIP path against 'the same computation' rather than against fp32 (random-init nets amplify bf16
    rounding by a lot).  No buffers are updated."""
    def r(t):
        return t.bfloat16().float()

    def cbr(t, ck, bk):
        acc = F.conv2d(r(t), r(sd[f"{ck}.weight"]), None, 1, 1)
        mean = acc.mean((0, 2, 3), keepdim=True)
        var = acc.var((0, 2, 3), unbiased=False, keepdim=True)
        invstd = torch.rsqrt(var + BN_EPS)
        scale = sd[f"{bk}.weight"].view(1, -1, 1, 1) * invstd
        shift = sd[f"{bk}.bias"].view(1, -1, 1, 1) - mean * scale
        return r(F.relu(r(acc) * scale + shift))

    def block(name, t):
        return cbr(cbr(t, f"{name}.conv.0", f"{name}.conv.1"), f"{name}.conv.3", f"{name}.conv.4")

    pool = lambda t: F.max_pool2d(t, 2, 2)  # noqa: E731
    e1 = block("Conv1", x)
    e2 = block("Conv2", pool(e1))
    e3 = block("Conv3", pool(e2))
    e4 = block("Conv4", pool(e3))
    d = block("Conv5", pool(e4))
    for lvl, skip in ((5, e4), (4, e3), (3, e2), (2, e1)):
        u = cbr(F.interpolate(d, scale_factor=2, mode="nearest"), f"Up{lvl}.up.1", f"Up{lvl}.up.2")
        d = block(f"Up_conv{lvl}", torch.cat((skip, u), 1))
    return F.conv2d(d, sd["DeConv_1x1.weight"], sd["DeConv_1x1.bias"])


class _RoundBF16(torch.autograd.Function):
    """bf16 rounding point of a stored tensor: the value is rounded going forward, its gradient going backward (the HIP path
    stores activation gradients in bf16 too); ``grad=False`` = a weight operand (its gradient stays fp32)."""

    @staticmethod
    def forward(ctx, x, grad):
        ctx.round_grad = grad
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return (g.bfloat16().float() if ctx.round_grad else g), None


def unet_forward_bf16_autograd(sd, x: Tensor, training: bool = True, update_stats: bool = False):
    """:func:`unet_forward` with the bf16 kernels' rounding points (see :func:`unet_forward_bf16_emulated`) as differentiable
    straight-through roundings, so that autograd yields the gradients the bf16 HIP step computes up to accumulation order and
    ReLU-mask flips: operands bf16, accumulation and batch statistics fp32, raw conv output / activations / their gradients
    stored in bf16, weight gradients and logits fp32.  Same return value as unet_forward.  No buffers are updated."""
    assert training
    r = _RoundBF16.apply
    f: "OrderedDict[str, Tensor]" = OrderedDict()

    def cbr(t, ck, bk):
        acc = F.conv2d(t, r(sd[f"{ck}.weight"], False), None, 1, 1)
        mean = acc.mean((0, 2, 3), keepdim=True)
        var = acc.var((0, 2, 3), unbiased=False, keepdim=True)
        scale = sd[f"{bk}.weight"].view(1, -1, 1, 1) * torch.rsqrt(var + BN_EPS)
        shift = sd[f"{bk}.bias"].view(1, -1, 1, 1) - mean * scale
        return r(F.relu(r(acc, True) * scale + shift), True)

    def block(name, t):
        return cbr(cbr(t, f"{name}.conv.0", f"{name}.conv.1"), f"{name}.conv.3", f"{name}.conv.4")

    pool = lambda t: F.max_pool2d(t, 2, 2)  # noqa: E731
    e1 = f["Conv1"] = block("Conv1", r(x, False))
    e2 = f["Conv2"] = block("Conv2", pool(e1))
    e3 = f["Conv3"] = block("Conv3", pool(e2))
    e4 = f["Conv4"] = block("Conv4", pool(e3))
    d = f["Conv5"] = block("Conv5", pool(e4))
    for lvl, skip in ((5, e4), (4, e3), (3, e2), (2, e1)):
        u = f[f"Up{lvl}"] = cbr(F.interpolate(d, scale_factor=2, mode="nearest"), f"Up{lvl}.up.1", f"Up{lvl}.up.2")
        d = f[f"Up_conv{lvl}"] = block(f"Up_conv{lvl}", torch.cat((skip, u), 1))
    logits = f["DeConv_1x1"] = F.conv2d(d, sd["DeConv_1x1.weight"], sd["DeConv_1x1.bias"])
    return logits, f
