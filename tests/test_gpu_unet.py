"""GPU parity: U-Net building blocks and the whole network (HIP through the C ABI) vs golden/oracle.

float32 mode uses the exact fp32 MFMA (k-ordered fma chain) and must match the reference's fp32
results to accumulation-order noise.  bfloat16 mode is checked two ways: (a) each layer against the
oracle evaluated on the SAME bf16-rounded operands (tight: only accumulation order differs), (b) the
whole network against the fp32 golden at bf16 tolerance.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import synth
from oracle import unet as OU

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda"


def nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


def make_bn(c, tag):
    return dict(weight=1 + 0.1 * T(synth.normal(tag + "/g", (c,))), bias=0.1 * T(synth.normal(tag + "/b", (c,))),
                running_mean=torch.zeros(c), running_var=torch.ones(c), nbt=torch.zeros((), dtype=torch.long))


def ref_layer(x, w, bn, training, pool):
    y = F.conv2d(x, w, None, 1, 1)
    rm, rv = bn["running_mean"].clone(), bn["running_var"].clone()
    y = F.batch_norm(y, rm, rv, bn["weight"], bn["bias"], training, 0.1, 1e-5)
    y = F.relu(y)
    return y, (F.max_pool2d(y, 2, 2) if pool else None), rm, rv


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", ["plain_pool", "concat", "upsample", "eval", "odd_size"])
def test_conv_bn_relu_layer(dtype, case):
    n, h, w = 3, 24, 40
    c0, c1, cout, ups0, pool, training = 16, 0, 32, 0, False, True
    if case == "plain_pool":
        pool = True
    elif case == "concat":
        c0, c1, cout = 32, 32, 16
    elif case == "upsample":
        c0, cout, ups0 = 64, 64, 1
    elif case == "eval":
        training = False
    elif case == "odd_size":
        h, w, c0, cout = 20, 36, 8, 32
    _layer_case(dtype, case, n, h, w, c0, c1, cout, ups0, pool, training)


# (n, h, w, c0, c1, cout, ups0, pool): every shape has >= 512 16x32 tiles and <= 32 input channels, i.e. goes to
# conv3x3_stream_kernel<COT, 32, NV, DU> (csrc/conv.hip conv_streams()), forward AND both dgrads; the six instantiations the
# 256^2 / 128^2 layers of the bench use are all here (16->16, 16->32, 32->32, upsampled 32->16, concat 16+16->16, stem 8->16)
STREAM_CASES = {
    "s16_16_pool": (4, 256, 256, 16, 0, 16, 0, True),
    "s16_32": (16, 128, 128, 16, 0, 32, 0, False),
    "s32_32_pool": (16, 128, 128, 32, 0, 32, 0, True),
    "s_up32_16": (4, 256, 256, 32, 0, 16, 1, False),
    "s_cat16_16_16": (4, 256, 256, 16, 16, 16, 0, False),
    "s_stem8_16": (4, 256, 256, 8, 0, 16, 0, False),
    "s_cat8_16_16": (6, 250, 230, 8, 16, 16, 0, False),      # ragged: H % 16 != 0, W % 32 != 0, 24 input channels
    "s24_32": (6, 250, 230, 24, 0, 32, 0, False),
}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", sorted(STREAM_CASES))
def test_conv_bn_relu_layer_streaming_bf16(case, dtype):
    """The persistent streaming conv (the kernel the bench's 256^2 / 128^2 layers run) against the oracle on the same bf16-rounded
    operands, same bounds as the generic kernel above: output within one bf16 ulp, gradients as test_conv_bn_relu_layer."""
    from miseg_amd import _cabi
    n, h, w, c0, c1, cout, ups0, pool = STREAM_CASES[case]
    assert c0 + c1 <= 32 and n * ((h + 15) // 16) * ((w + 31) // 32) >= 512     # conv_streams() of csrc/conv.hip
    # the streaming kernel hands BatchNorm one partial per persistent block (<= 512), the generic one a partial per tile
    tiles = n * ((h + 15) // 16) * ((w + 31) // 32)
    assert _cabi.query("miseg_conv3x3_stats_parts", _cabi.BF16 if dtype == torch.bfloat16 else _cabi.F16, c0 + c1, n, h, w) == min(tiles, 512)
    _layer_case(dtype, case, n, h, w, c0, c1, cout, ups0, pool, True)


def _layer_case(dtype, case, n, h, w, c0, c1, cout, ups0, pool, training):
    from miseg_amd import unet_ops
    half = dtype in (torch.bfloat16, torch.float16)
    rnd = (lambda t: t.to(dtype).float()) if half else (lambda t: t)
    x0 = rnd(T(synth.normal(f"layer/{case}/x0", (n, c0, h >> ups0, w >> ups0))))
    x1 = rnd(T(synth.normal(f"layer/{case}/x1", (n, c1, h, w)))) if c1 else None
    wt = T(synth.normal(f"layer/{case}/w", (cout, c0 + c1, 3, 3), scale=(2.0 / ((c0 + c1) * 9)) ** 0.5))
    bn = make_bn(cout, f"layer/{case}/bn")
    if not training:
        bn["running_mean"] = 0.1 * T(synth.normal(f"layer/{case}/rm", (cout,)))
        bn["running_var"] = 1 + 0.2 * T(synth.uniform(f"layer/{case}/rv", (cout,)))
    cot = T(synth.normal(f"layer/{case}/cot", (n, cout, h, w)))
    cotp = T(synth.normal(f"layer/{case}/cotp", (n, cout, h // 2, w // 2)))
    # ---- oracle on the operands as the kernel sees them
    x0r, wr = x0.clone().requires_grad_(True), rnd(wt).clone().requires_grad_(True)
    x1r = x1.clone().requires_grad_(True) if c1 else None
    gr, br = bn["weight"].clone().requires_grad_(True), bn["bias"].clone().requires_grad_(True)
    xin = F.interpolate(x0r, scale_factor=2, mode="nearest") if ups0 else x0r
    xin = torch.cat((xin, x1r), 1) if c1 else xin
    yr, pr, rm, rv = ref_layer(xin, wr, dict(bn, weight=gr, bias=br), training, pool)
    obj = (yr * cot).sum() + ((pr * cotp).sum() if pool else 0)
    obj.backward()
    # ---- HIP
    x0d = nhwc(x0.to(DEV).to(dtype)).requires_grad_(True)
    x1d = nhwc(x1.to(DEV).to(dtype)).requires_grad_(True) if c1 else None
    wd = wt.to(DEV).requires_grad_(True)
    gd, bd = bn["weight"].to(DEV).requires_grad_(True), bn["bias"].to(DEV).requires_grad_(True)
    rmd, rvd, nbt = bn["running_mean"].to(DEV), bn["running_var"].to(DEV), bn["nbt"].to(DEV)
    y, p = unet_ops.conv_bn_relu(x0d, x1d, wd, gd, bd, rmd, rvd, nbt, training, ups0, 0, pool)
    # 16-bit modes: one ulp of the stored output (bf16 2^-7 at magnitude 2, IEEE half 2^-10)
    tol = {torch.float32: dict(rtol=2e-5, atol=2e-5), torch.bfloat16: dict(rtol=1.6e-2, atol=1.6e-2),
           torch.float16: dict(rtol=2e-3, atol=2e-3)}[dtype]
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), yr.detach().numpy(), **tol)
    if pool:
        np.testing.assert_allclose(p.detach().float().cpu().numpy(), pr.detach().numpy(), **tol)
    if training:
        np.testing.assert_allclose(rmd.cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(rvd.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-6)
        assert int(nbt) == 1
    obj_d = (y.float() * cot.to(DEV)).sum() + ((p.float() * cotp.to(DEV)).sum() if pool else 0)
    obj_d.backward()
    gt = {torch.float32: dict(rtol=2e-4, atol=2e-4), torch.bfloat16: dict(rtol=5e-2, atol=8e-2),
          torch.float16: dict(rtol=8e-3, atol=1e-2)}[dtype]

    def close(a, b, name):
        a, b = a.float().cpu().numpy(), b.numpy()
        bad = np.abs(a - b) > gt["atol"] * max(1.0, float(np.abs(b).max())) + gt["rtol"] * np.abs(b)
        # bf16 + max-pool: rounding y to bf16 creates ties inside 2x2 windows that fp32 does not have, so a
        # handful of pooled gradients are routed to a different (equal-valued) pixel than the fp32 oracle picks.
        # bf16 in general: a stored activation that rounds to exactly 0 flips its ReLU mask vs the fp32 oracle.
        # IEEE half: fewer flips, but its 8x tighter bound no longer hides the small ones (measured 6e-4 of gx0 at s16_32)
        allowed = (5e-3 if (pool and name == "gx0") else (2e-3 if dtype == torch.float16 else 5e-4)) if half else 0.0
        assert bad.mean() <= allowed, (name, float(bad.mean()), float(np.abs(a - b).max()))
    close(wd.grad, wr.grad, "gw")
    close(gd.grad, gr.grad, "ggamma")
    close(bd.grad, br.grad, "gbeta")
    close(x0d.grad, x0r.grad, "gx0")
    if c1:
        close(x1d.grad, x1r.grad, "gx1")


def load_unet(dtype, seed=3):
    from contrastyou.arch import UNet
    net = UNet(input_dim=1, num_classes=4, compute_dtype=dtype)
    net.load_state_dict(OU.init_state(1, 4, seed=seed))
    return net.to(DEV)


def test_unet_fp32_vs_golden(golden):
    """Whole network in exact-fp32 mode vs the reference's own outputs (logits, taps, grads, BN buffers)."""
    g = golden("unet")
    net = load_unet(torch.float32)
    x = T(synth.uniform("unet/x64", (3, 1, 64, 64))).to(DEV)
    wgt = T(synth.normal("unet/w64", (3, 4, 64, 64))).to(DEV)
    net.train()
    logits, enc, dec = net(x, return_features=True)
    assert logits.shape == (3, 4, 64, 64) and logits.dtype == torch.float32
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["train64/logits"], rtol=2e-4, atol=2e-4)
    for name, f in list(zip(("Conv5", "Conv4", "Conv3", "Conv2", "Conv1"), enc)) + list(zip(("Up_conv5", "Up_conv4", "Up_conv3", "Up_conv2"), dec)):
        synth.check_fingerprint(f.detach().float().cpu().contiguous().numpy(), synth.fp_unpack(g, f"train64/feat/{name}"),
                                f"train64/feat/{name}", rtol=2e-4, atol=2e-4)
    (logits * wgt).sum().backward()
    # End-to-end gradients are limited by ReLU-mask flips, not by kernel accuracy: the HIP forward differs from
    # torch-CPU by ~1e-6..1e-5 (summation order), so roughly one activation per deep layer lands on the other side
    # of zero, and ONE flipped element moves a BatchNorm-bias gradient (a sum over ~3k pixels of mixed sign) by
    # ~1e-2 relative; everything upstream inherits it.  (The reference's own fp32 run deviates from an fp64 run
    # by up to 5e-3 for the same reason.)  Kernel-level accuracy is pinned tightly by test_conv_bn_relu_layer.
    worst = {}
    for k, p in net.named_parameters():
        fp = synth.fp_unpack(g, f"train64/grad/{k}")
        got = p.grad.cpu().numpy().reshape(-1).astype(np.float64)[synth.sample_index(p.numel(), f"train64/grad/{k}")]
        worst[k] = float(np.linalg.norm(got - fp["sample"]) / (np.linalg.norm(fp["sample"]) + 1e-30))
    bad = {k: v for k, v in worst.items() if v > 5e-2}
    assert not bad, bad
    # The last layers sit behind at most two ReLUs, so most of their gradients see no flip at all and must be tight.
    # (Not all: this input has an Up_conv2 pre-activation of 3.9e-6 at scale 0.78 -- scratch analysis with the fp64
    # oracle -- which lands on either side of zero depending on the fp32 summation order of the BN statistics.)
    tail = sorted(worst[k] for k in worst if k.startswith(("Up_conv2", "DeConv")))
    assert tail[len(tail) // 2] < 1e-4 and tail[0] < 1e-5, tail
    for k, v in net.state_dict().items():
        if "running" in k or "num_batches" in k:
            np.testing.assert_allclose(v.cpu().numpy(), g[f"train64/after/{k}"], rtol=1e-4, atol=1e-5)
    net.eval()
    with torch.no_grad():
        ev = net(x)
    np.testing.assert_allclose(ev.cpu().numpy(), g["eval64/logits"], rtol=2e-4, atol=2e-4)


def test_unet_256_fp32_and_bf16(golden):
    g = golden("unet")
    x = T(synth.uniform("unet/x256", (2, 1, 256, 256))).to(DEV)
    net = load_unet(torch.float32).train()
    with torch.no_grad():
        l32 = net(x)
    synth.check_fingerprint(l32.cpu().contiguous().numpy(), synth.fp_unpack(g, "train256/logits"), "train256/logits", rtol=3e-4, atol=3e-4)
    netb = load_unet(torch.bfloat16).train()
    with torch.no_grad():
        lb = netb(x)
        emu = OU.unet_forward_bf16_emulated(OU.init_state(1, 4, seed=3), x.cpu()).to(DEV)

    def rel_rms(a, b):
        return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
    # (a) against the SAME computation (bf16 rounding points emulated on the CPU): only accumulation order differs
    assert rel_rms(lb, emu) < 8e-2, rel_rms(lb, emu)   # rare bf16 rounding flips, amplified by the random-init net
    assert (lb.argmax(1) == emu.argmax(1)).float().mean().item() > 0.96
    # (b) against fp32: a random-init 23-layer net amplifies bf16 operand rounding to ~17 % RMS (the CPU emulation
    # shows the same 0.166), so this bound only guards against gross errors
    assert rel_rms(lb, l32) < 0.25, rel_rms(lb, l32)
    assert abs(rel_rms(lb, l32) - rel_rms(emu, l32)) < 0.04
    # (c) IEEE-half storage (the f16 build of the same kernels): 3 more mantissa bits -> ~8x closer to fp32 than bf16 is
    with torch.no_grad():
        lh = load_unet(torch.float16).train()(x)
    assert rel_rms(lh, l32) < 0.3 * rel_rms(lb, l32), (rel_rms(lh, l32), rel_rms(lb, l32))
    assert (lh.argmax(1) == l32.argmax(1)).float().mean().item() > 0.98


def test_unet_hooks_and_state_dict_keys():
    net = load_unet(torch.float32).train()
    ref_keys = list(OU.init_state(1, 4).keys())
    assert list(net.state_dict().keys()) == ref_keys
    seen = {}
    handles = [getattr(net, n).register_forward_hook(lambda m, i, o, n=n: seen.__setitem__(n, o)) for n in net.component_names]
    x = torch.rand(2, 1, 32, 32, device=DEV)
    logits, enc, dec = net(x, return_features=True)
    for h in handles:
        h.remove()
    assert set(seen) == set(net.component_names)
    assert seen["Conv5"] is enc[0] and seen["Up_conv2"] is dec[-1] and seen["DeConv_1x1"] is logits
    assert seen["Up_conv3"].shape == (2, 32, 16, 16)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(4, 256, 256, 16, 16), (16, 128, 128, 32, 32), (48, 16, 16, 256, 256), (6, 50, 46, 24, 32), (48, 64, 64, 64, 64)])
def test_bn_statistics_finished_by_the_last_block_equal_the_separate_finalize(shape, dtype, monkeypatch):
    """miseg_conv3x3_bn_fwd / miseg_bn_relu_bwd_sync (the block that arrives last sums the partial rows and writes the coefficients)
    against the two-launch form (miseg_bn_finalize / bn_bwd_finalize_kernel): the partial sums are the same numbers, only their
    summation order differs -> 1e-6 relative on every statistic and gradient; run twice to prove the counter resets itself.
    The last shape is above the fusable limit on the forward side (separate finalize) and fused on the backward side."""
    from miseg_amd import unet_ops
    n, h, w, cin, cout = shape
    x = nhwc(T(synth.normal(f"bnfin/{shape}/x", (n, cin, h, w))).to(DEV).to(dtype))
    wt = T(synth.normal(f"bnfin/{shape}/w", (cout, cin, 3, 3), scale=(2.0 / (cin * 9)) ** 0.5)).to(DEV)
    cot = T(synth.normal(f"bnfin/{shape}/cot", (n, cout, h, w))).to(DEV)

    def run(enabled):
        monkeypatch.setattr(unet_ops.SYNC_COUNTERS, "enabled", enabled)
        bn = {k: v.to(DEV) for k, v in make_bn(cout, f"bnfin/{shape}/bn").items()}
        xd, wd = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
        gd, bd = bn["weight"].clone().requires_grad_(True), bn["bias"].clone().requires_grad_(True)
        y, _ = unet_ops.conv_bn_relu(xd, None, wd, gd, bd, bn["running_mean"], bn["running_var"], bn["nbt"], True, 0, 0, False)
        (y.float() * cot).sum().backward()
        return [t.detach().float().cpu() for t in (y, bn["running_mean"], bn["running_var"], gd.grad, bd.grad, xd.grad, wd.grad)] + [int(bn["nbt"])]

    ref = run(False)
    for _ in range(2):
        got = run(True)
        assert got[-1] == ref[-1] == 1
        for a, b, name in zip(got[:-1], ref[:-1], ("y", "running_mean", "running_var", "ggamma", "gbeta", "gx", "gw")):
            tol = 1e-6 if dtype == torch.float32 or name.startswith(("running", "gg", "gb")) else 1.6e-2   # 16-bit outputs: one ulp
            err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)
            assert err <= tol, (name, err)
    if unet_ops.SYNC_COUNTERS._pool:
        assert all(int(p.abs().sum()) == 0 for p in unet_ops.SYNC_COUNTERS._pool.values())      # every counter is back at zero


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(4, 256, 256, 16, 16), (16, 128, 128, 32, 32), (48, 16, 16, 256, 256), (6, 50, 46, 24, 32), (48, 64, 64, 64, 64),
                                   (8, 32, 32, 128, 128)])
def test_bn_statistics_as_fixed_point_accumulators_equal_the_finalize_launch(shape, dtype, monkeypatch):
    """miseg_conv3x3_fwd_acc + miseg_bn_relu_fwd_acc (the shipped training path: every block of the convolution adds its channel sums to
    one int64 accumulator, the apply kernel finishes the statistics in its prologue; and miseg_bn_relu_bwd_dual_acc, the opt-in twin for
    the backward's sums) against miseg_conv3x3_fwd + miseg_bn_finalize + miseg_bn_relu_fwd (and the backward with its finalize launch): same per-block sums, exacter total -> 2e-6 relative on every statistic and gradient (16-bit outputs: one ulp);
    streaming, persistent tiled and per-tile kernels, pooled and unpooled apply kernels.  Two runs of the accumulator path are
    IDENTICAL bit for bit (integer adds do not care about the order the blocks arrive in)."""
    from miseg_amd import unet_ops
    n, h, w, cin, cout = shape
    x = nhwc(T(synth.normal(f"bnacc/{shape}/x", (n, cin, h, w))).to(DEV).to(dtype))
    wt = T(synth.normal(f"bnacc/{shape}/w", (cout, cin, 3, 3), scale=(2.0 / (cin * 9)) ** 0.5)).to(DEV)
    pool = h % 2 == 0 and w % 2 == 0
    cot = T(synth.normal(f"bnacc/{shape}/cot", (n, cout, h, w))).to(DEV)
    monkeypatch.setattr(unet_ops.SYNC_COUNTERS, "enabled", False)

    def run(acc):
        monkeypatch.setattr(unet_ops, "_BN_ACC", acc)
        monkeypatch.setattr(unet_ops, "_BN_ACC_BWD", acc)         # (the backward's two-tier accumulators: opt-in, same parity bar)
        bn = {k: v.to(DEV) for k, v in make_bn(cout, f"bnacc/{shape}/bn").items()}
        xd, wd = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
        gd, bd = bn["weight"].clone().requires_grad_(True), bn["bias"].clone().requires_grad_(True)
        y, yp = unet_ops.conv_bn_relu(xd, None, wd, gd, bd, bn["running_mean"], bn["running_var"], bn["nbt"], True, 0, 0, pool)
        loss = (y.float() * cot).sum() + (yp.float().sum() if pool else 0.0)
        loss.backward()
        outs = [y] + ([yp] if pool else []) + [bn["running_mean"], bn["running_var"], gd.grad, bd.grad, xd.grad, wd.grad]
        return [t.detach().float().cpu() for t in outs] + [int(bn["nbt"])]

    ref, got, again = run(False), run(True), run(True)
    names = ["y"] + (["pooled"] if pool else []) + ["running_mean", "running_var", "ggamma", "gbeta", "gx", "gw"]
    assert got[-1] == ref[-1] == 1
    for a, b, c, name in zip(got[:-1], ref[:-1], again[:-1], names):
        assert torch.equal(a, c), name
        tol = 2e-6 if dtype == torch.float32 or name.startswith(("running", "gg", "gb")) else 1.6e-2
        err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)
        assert err <= tol, (name, err)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(4, 64, 96, 16), (3, 50, 46, 16), (2, 256, 256, 16), (5, 33, 20, 32)])
def test_stem_kernels_equal_the_mfma_path_and_the_fp32_reference(shape, dtype, monkeypatch):
    """The stem's own kernels (miseg_conv3x3_stem_fwd / _wgrad: a pixel and four output channels per thread, no matrix cores) against the
    streaming MFMA convolution + tiled weight gradient on the padded channel vector, and against torch in fp32 on the same 16-bit operands:
    same products, fp32 sums in another order -> outputs within one 16-bit ulp, statistics 2e-6, parameter gradients 5e-3 relative L2;
    training and evaluation mode, ragged sizes."""
    from miseg_amd import unet_ops
    n, h, w, cout = shape
    img = T(synth.normal(f"stem/{shape}/x", (n, 1, h, w))).to(DEV).to(dtype)
    xpad = nhwc(torch.cat([img, torch.zeros(n, 7, h, w, device=DEV, dtype=dtype)], 1))
    wt = T(synth.normal(f"stem/{shape}/w", (cout, 1, 3, 3), scale=(2.0 / 9) ** 0.5)).to(DEV)
    cot = T(synth.normal(f"stem/{shape}/cot", (n, cout, h, w))).to(DEV)

    img32 = img.float().contiguous()           # (16-bit values: the descriptor form rounds them to themselves)

    def run(stem, training=True, descriptor=False):
        monkeypatch.setattr(unet_ops, "_STEM_KERNELS", stem)
        bn = {k: v.to(DEV) for k, v in make_bn(cout, f"stem/{shape}/bn").items()}
        wd = wt.clone().requires_grad_(True)
        gd, bd = bn["weight"].clone().requires_grad_(True), bn["bias"].clone().requires_grad_(True)
        xin = unet_ops.stem_input(img32, dtype) if descriptor else xpad        # descriptor: what the network passes (the fp32 image rides along)
        assert (getattr(xin, "_miseg_stem_f32", None) is not None) == descriptor
        y, _ = unet_ops.conv_bn_relu(xin, None, wd, gd, bd, bn["running_mean"], bn["running_var"], bn["nbt"], training, 0, 0, False)
        (y.float() * cot).sum().backward()
        return [t.detach().float().cpu() for t in (y, bn["running_mean"], bn["running_var"], gd.grad, bd.grad, wd.grad)]

    for training in (True, False):
        got, ref, dsc = run(True, training), run(False, training), run(True, training, descriptor=True)
        for a, d in zip(got, dsc):
            assert torch.equal(a, d)            # the stem kernels on the padded operand and on the fp32 image: the same numbers
        for a, b, name in zip(got, ref, ("y", "running_mean", "running_var", "ggamma", "gbeta", "gw")):
            if name == "y":
                assert (a - b).abs().max().item() <= 1.6e-2 * b.abs().max().item(), name
            elif name.startswith("running"):
                assert (a - b).abs().max().item() <= 2e-6 * (b.abs().max().item() + 1.0), name
            else:
                assert (a - b).norm().item() <= 5e-3 * (b.norm().item() + 1e-30), (name, (a - b).norm().item() / b.norm().item())
    # fp32 torch on the same 16-bit operands (weights rounded as the kernels round them)
    w16 = wt.to(dtype).float()
    y32, _, rm, rv = ref_layer(img.float(), w16, {k: v.to(DEV) for k, v in make_bn(cout, f"stem/{shape}/bn").items()}, True, False)
    got = run(True, True)
    assert (got[0] - y32.cpu()).abs().max().item() <= 1.6e-2 * y32.abs().max().item()
    assert (got[1] - rm.cpu()).abs().max().item() <= 1e-5 and (got[2] - rv.cpu()).abs().max().item() <= 1e-5 * float(rv.abs().max())


def test_bn_statistics_accumulator_poisoned_by_a_non_finite_activation():
    """A non-finite (or absurdly large) block sum must not turn into a plausible statistic: the accumulator is poisoned and the batch
    statistics come out NaN, as with the float path (the kernels' ReLU is fmaxf(v, 0): the activations themselves read 0 either way)."""
    from miseg_amd import unet_ops
    n, h, w, c = 2, 16, 32, 16
    x = nhwc(torch.randn(n, c, h, w, device=DEV).to(torch.bfloat16))
    x[0, 3, 5, 7] = float("inf")
    wt = (torch.randn(c, c, 3, 3, device=DEV) * 0.1)
    bn = {k: v.to(DEV) for k, v in make_bn(c, "bnacc/poison").items()}
    assert unet_ops._BN_ACC
    y, _ = unet_ops.conv_bn_relu(x, None, wt, bn["weight"], bn["bias"], bn["running_mean"], bn["running_var"], bn["nbt"], True, 0, 0, False)
    assert not bool(torch.isfinite(bn["running_var"]).all()) and not bool(torch.isfinite(bn["running_mean"]).all())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(4, 256, 256, 32, 16), (16, 128, 128, 32, 32), (6, 250, 230, 16, 32),
                                   (8, 64, 64, 128, 64), (8, 32, 32, 256, 128), (3, 20, 36, 64, 64)])     # the last three: tiled kernel, 64-channel slices
def test_upsampled_source_gradient_pooled_in_the_dgrad_epilogue(shape, dtype, monkeypatch):
    """Backward of a convolution that reads its input through the nearest x2 upsample (ref unet.py:32): the source gradient is the
    2x2 sum-pool of the full-resolution data gradient.  `miseg_conv3x3_fwd_sumpool` pools the fp32 accumulators in the epilogue; the
    two-launch form (conv3x3 dgrad -> miseg_sumpool2x2) rounds every full-resolution value to 16 bits first.  Both against the fp32
    oracle: the fused form must be at least as close, and the two must agree to the two-rounding bound (incl. a ragged shape)."""
    from miseg_amd import unet_ops
    n, h, w, cin, cout = shape
    x = nhwc(T(synth.normal(f"upsd/{shape}/x", (n, cin, h // 2, w // 2))).to(DEV).to(dtype))
    wt = T(synth.normal(f"upsd/{shape}/w", (cout, cin, 3, 3), scale=(2.0 / (cin * 9)) ** 0.5))
    cot = T(synth.normal(f"upsd/{shape}/cot", (n, cout, h, w)))
    bn = make_bn(cout, f"upsd/{shape}/bn")

    def run(fused):
        monkeypatch.setattr(unet_ops, "_FUSE_UPS_DGRAD", fused)
        b = {k: v.clone().to(DEV) for k, v in bn.items()}
        xd = x.clone().requires_grad_(True)
        y, _ = unet_ops.conv_bn_relu(xd, None, wt.to(DEV).requires_grad_(True), b["weight"].requires_grad_(True), b["bias"].requires_grad_(True),
                                     b["running_mean"], b["running_var"], b["nbt"], True, 1, 0, False)
        (y.float() * cot.to(DEV)).sum().backward()
        return xd.grad.float().cpu()

    fused, two = run(True), run(False)
    xr = x.float().cpu().requires_grad_(True)
    yr, _, _, _ = ref_layer(F.interpolate(xr, scale_factor=2, mode="nearest"), wt.to(dtype).float(), bn, True, False)
    (yr * cot).sum().backward()
    scale = float(xr.grad.abs().max())
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert float((fused - two).abs().max()) <= 4 * ulp * scale
    e_fused, e_two = float((fused - xr.grad).abs().mean()), float((two - xr.grad).abs().mean())
    assert e_fused <= 1.05 * e_two + 1e-9, (e_fused, e_two)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("n,h,w,cin,cout", [(48, 256, 256, 16, 16), (48, 128, 128, 32, 32), (48, 64, 64, 64, 64), (48, 16, 16, 256, 256), (5, 250, 230, 24, 32)])
def test_conv3x3_exact_properties_at_full_size(n, h, w, cin, cout, dtype):
    """The convolution entry points at the bench's full layer sizes (streaming and tiled kernels, every storage type), checked by
    properties that hold EXACTLY: (a) a centre-tap identity kernel returns the input bit for bit; (b) all-ones weights on an all-ones
    input give (valid taps) x Cin at every pixel -- 9 Cin inside, 6 Cin on edges, 4 Cin in corners (zero padding, ref unet.py:15);
    (c) the pooled-output form (`miseg_conv3x3_fwd_sumpool`, where supported) gives the 2 x 2 sums of (b)."""
    from miseg_amd import _cabi, unet_ops
    from miseg_amd.ops import _DT
    dt = _DT[dtype]
    st = torch.cuda.current_stream().cuda_stream
    x = nhwc(T(synth.normal(f"convprop/{n}/{h}/{cin}", (n, cin, h, w))).to(DEV).to(dtype))
    # (a) identity
    if cin == cout:
        wid = torch.zeros(cout, cin, 3, 3, device=DEV)
        wid[torch.arange(cout), torch.arange(cin), 1, 1] = 1.0
        out = torch.empty_like(x)
        _cabi.call("miseg_conv3x3_fwd", st, dt, x.data_ptr(), cin, 0, None, 0, 0, n, h, w, unet_ops._pack_now(wid, dtype, 0, 0, cin).data_ptr(), cout,
                   out.data_ptr(), None)
        assert torch.equal(out, x)
    # (b) tap counts
    ones = nhwc(torch.ones(n, cin, h, w, device=DEV, dtype=dtype))
    wone = torch.ones(cout, cin, 3, 3, device=DEV)
    packed = unet_ops._pack_now(wone, dtype, 0, 0, cin)
    out = nhwc(torch.empty(n, cout, h, w, device=DEV, dtype=dtype))
    _cabi.call("miseg_conv3x3_fwd", st, dt, ones.data_ptr(), cin, 0, None, 0, 0, n, h, w, packed.data_ptr(), cout, out.data_ptr(), None)
    r = torch.full((h,), 3.0); r[0] = r[-1] = 2.0
    c = torch.full((w,), 3.0); c[0] = c[-1] = 2.0
    taps = (torch.outer(r, c) * cin).to(DEV)
    assert torch.equal(out.float(), taps.expand(n, cout, h, w))                     # integers <= 2304: exact in bf16 / half / fp32 sums
    # (c) pooled output
    if dtype != torch.float32 and _cabi.query("miseg_conv3x3_fwd_sumpool_supported", dt, cin, n, h, w, cout):
        pooled = nhwc(torch.empty(n, cout, h // 2, w // 2, device=DEV, dtype=dtype))
        _cabi.call("miseg_conv3x3_fwd_sumpool", st, dt, ones.data_ptr(), cin, n, h, w, packed.data_ptr(), cout, pooled.data_ptr())
        want = F.avg_pool2d(taps[None, None], 2)[0, 0] * 4
        if float(want.max()) <= 2048 or dtype == torch.float16 and float(want.max()) <= 2048:
            assert torch.equal(pooled.float(), want.expand(n, cout, h // 2, w // 2))
        else:                                                                        # sums above the 16-bit type's exact-integer range: one rounding
            assert float(((pooled.float() - want) / want).abs().max()) <= 2.0 ** -8


# ---- BatchNorm backward folded into the convolutions (unet_ops._backward_fused, csrc/conv.hip BNL / RED): a chain of layers, so that
# the second layer's data-gradient kernel forms graw in its loader AND takes the first layer's backward sums in its epilogue.
# (n, h, w, channel chain, upsample before layer index or None, concat the first activation into layer index or None)
CHAIN_CASES = {
    "tiled_64": (6, 64, 64, (64, 64, 64), None, None),             # conv3x3_kernel<.., BNL, RED> (64 channels: not a streaming shape)
    "tiled_ragged": (3, 20, 36, (16, 32, 16), None, None),          # partial tiles
    "stream_16": (4, 256, 256, (16, 16, 16), None, None),           # conv3x3_stream_kernel<16, 32, 2, .., BNL, RED, 4>
    "stream_32": (16, 128, 128, (32, 32, 32), None, None),          # <32, 32, 4, .., BNL, RED, 2> (8-row tiles)
    "stream_32_16": (16, 128, 128, (16, 32, 32), None, None),       # data gradient 32 -> 16 channels: <16, 32, 4, .., 2>
    "stream_ragged": (6, 250, 230, (16, 32, 16), None, None),
    "stream_up": (4, 256, 256, (32, 16, 16), 1, None),              # pooled data gradient of the upsampled layer, BN loader
    "stream_cat": (4, 256, 256, (16, 16, 16, 16), None, 2),         # concat: the first activation has two consumers -> no hand-over
}


def _chain(dtype, case, fuse, monkeypatch):
    from miseg_amd import unet_ops
    monkeypatch.setattr(unet_ops, "_FUSE_BN_BWD", fuse)
    n, h, w, chain, up_at, cat_at = CHAIN_CASES[case]
    hin, win = (h // 2, w // 2) if up_at == 0 else (h, w)
    x = nhwc(T(synth.normal(f"chain/{case}/x", (n, chain[0], hin, win))).to(DEV).to(dtype)).requires_grad_(True)
    params, acts = [], []
    cur, cin = x, chain[0]
    for i, cout in enumerate(chain[1:]):
        x1 = acts[0] if cat_at == i else None
        ctot = cin + (x1.shape[1] if x1 is not None else 0)
        wt = T(synth.normal(f"chain/{case}/w{i}", (cout, ctot, 3, 3), scale=(2.0 / (ctot * 9)) ** 0.5)).to(DEV).requires_grad_(True)
        bn = make_bn(cout, f"chain/{case}/bn{i}")
        g, b = bn["weight"].to(DEV).requires_grad_(True), bn["bias"].to(DEV).requires_grad_(True)
        cur, _ = unet_ops.conv_bn_relu(cur, x1, wt, g, b, bn["running_mean"].to(DEV), bn["running_var"].to(DEV), bn["nbt"].to(DEV), True,
                                       1 if up_at == i else 0, 0, False)
        params += [wt, g, b]
        acts.append(cur)
        cin = cout
    cot = T(synth.normal(f"chain/{case}/cot", tuple(cur.shape))).to(DEV)
    (cur.float() * cot).sum().backward()
    return [x.grad] + [p.grad for p in params]


def _chain_reference(dtype, case):
    n, h, w, chain, up_at, cat_at = CHAIN_CASES[case]
    rnd = (lambda t: t.to(dtype).float()) if dtype != torch.float32 else (lambda t: t)
    hin, win = (h // 2, w // 2) if up_at == 0 else (h, w)
    x = rnd(T(synth.normal(f"chain/{case}/x", (n, chain[0], hin, win)))).to(DEV).requires_grad_(True)
    params, acts = [], []
    cur, cin = x, chain[0]
    for i, cout in enumerate(chain[1:]):
        x1 = acts[0] if cat_at == i else None
        ctot = cin + (x1.shape[1] if x1 is not None else 0)
        wt = rnd(T(synth.normal(f"chain/{case}/w{i}", (cout, ctot, 3, 3), scale=(2.0 / (ctot * 9)) ** 0.5))).to(DEV).requires_grad_(True)
        bn = make_bn(cout, f"chain/{case}/bn{i}")
        g, b = bn["weight"].to(DEV).requires_grad_(True), bn["bias"].to(DEV).requires_grad_(True)
        xin = F.interpolate(cur, scale_factor=2, mode="nearest") if up_at == i else cur
        xin = torch.cat((xin, x1), 1) if x1 is not None else xin
        cur = F.relu(F.batch_norm(F.conv2d(xin, wt, None, 1, 1), None, None, g, b, True, 0.1, 1e-5))
        params += [wt, g, b]
        acts.append(cur)
        cin = cout
    cot = T(synth.normal(f"chain/{case}/cot", tuple(cur.shape))).to(DEV)
    (cur * cot).sum().backward()
    return [x.grad] + [p.grad for p in params]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("case", sorted(CHAIN_CASES))
def test_bn_backward_folded_into_the_convolutions(case, dtype, monkeypatch):
    """Fused against the three-pass form (same arithmetic up to fp32 rounding order before the one rounding to the storage type) and
    against torch's fp32 autograd of the same chain."""
    if dtype == torch.float32 and case.startswith("stream"):
        n, h, w, chain, up_at, cat_at = CHAIN_CASES[case]
        if n * h * w > 6 * 250 * 230:
            pytest.skip("fp32 runs the tiled kernel; one large case is enough")
    from miseg_amd import _cabi
    calls = []
    real_call = _cabi.call

    def spy(name, *a, **k):
        calls.append((name, a))
        return real_call(name, *a, **k)
    monkeypatch.setattr("miseg_amd.unet_ops.call", spy)
    fused = _chain(dtype, case, True, monkeypatch)
    names = [c[0] for c in calls]
    n_layers = len(CHAIN_CASES[case][3]) - 1
    assert names.count("miseg_conv3x3_wgrad_bn") == n_layers and "miseg_bn_relu_bwd_sync" not in names
    # the epilogue reduce happened wherever an activation had exactly one consumer: its stats call came with external parts
    ext = [a for nme, a in calls if nme == "miseg_bn_relu_bwd_stats" and a[14] is not None]
    up_at, cat_at = CHAIN_CASES[case][4:6]
    assert len(ext) == n_layers - 1 - (cat_at is not None) - (up_at is not None and up_at >= 1), (len(ext), names)
    calls.clear()
    plain = _chain(dtype, case, False, monkeypatch)
    assert "miseg_conv3x3_wgrad_bn" not in [c[0] for c in calls]
    ref = _chain_reference(dtype, case)
    half = dtype != torch.float32
    for i, (a, b, r) in enumerate(zip(fused, plain, ref)):
        a, b, r = a.float(), b.float(), r.float()
        scale = float(r.abs().max()) + 1e-12
        # fused vs three-pass: graw differs by fp32 rounding order only, i.e. by at most one ulp of the storage type on a few elements
        d_fp = float((a - b).abs().max()) / scale
        assert d_fp <= (3e-2 if dtype == torch.bfloat16 else 4e-3 if half else 2e-5), (case, i, d_fp)
        rel_l2 = float((a - b).norm() / (b.norm() + 1e-20))
        assert rel_l2 <= (2e-3 if dtype == torch.bfloat16 else 3e-4 if half else 1e-5), (case, i, rel_l2)   # fp32: summation order (measured 4.5e-6)
        # vs the fp32 reference on the same rounded operands: the bounds of _layer_case, as a relative L2 (mask flips are local)
        rel_ref = float((a - r).norm() / (r.norm() + 1e-20))
        # (measured: bf16 <= 0.066 -- the gamma gradients, a cancelling sum --, half <= 0.015, fp32 <= 1.1e-3: a few masks at |y| ~ 1e-7)
        assert rel_ref <= (1e-1 if dtype == torch.bfloat16 else 3e-2 if half else 3e-3), (case, i, rel_ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("n,h,w,c0,c1,cout", [(4, 256, 256, 16, 16, 16), (16, 128, 128, 32, 32, 32), (48, 64, 64, 64, 64, 64),
                                              (48, 32, 32, 128, 128, 128), (6, 250, 230, 16, 8, 16), (3, 20, 36, 32, 12, 16)])
def test_concat_data_gradient_in_one_launch_equals_the_two_sliced_launches(n, h, w, c0, c1, cout, dtype, monkeypatch):
    """miseg_conv3x3_dgrad_dual (one launch, two destinations) against the per-source launches: the same products in the same order,
    so the two gradients must agree bit for bit -- tiled and streaming kernels, ragged tiles, a second source of 8 / 12 channels."""
    from miseg_amd import unet_ops
    if dtype == torch.float32 and n * h * w > 48 * 64 * 64:
        pytest.skip("fp32 runs the tiled kernel; the smaller cases cover it")
    if dtype != torch.float32 and (c1 % 8 or cout % 8):
        pytest.skip("16-bit storage needs whole 16-byte channel vectors")
    outs = []
    for dual in (True, False):
        monkeypatch.setattr(unet_ops, "_DUAL_DGRAD", dual)
        names = []
        real = unet_ops.call
        monkeypatch.setattr(unet_ops, "call", lambda name, *a, **k: (names.append(name), real(name, *a, **k))[1])
        x0 = nhwc(T(synth.normal("dual/x0", (n, c0, h, w))).to(DEV).to(dtype)).requires_grad_(True)
        x1 = nhwc(T(synth.normal("dual/x1", (n, c1, h, w))).to(DEV).to(dtype)).requires_grad_(True)
        wt = T(synth.normal("dual/w", (cout, c0 + c1, 3, 3), scale=(2.0 / ((c0 + c1) * 9)) ** 0.5)).to(DEV).requires_grad_(True)
        bn = make_bn(cout, "dual/bn")
        y, _ = unet_ops.conv_bn_relu(x0, x1, wt, bn["weight"].to(DEV).requires_grad_(True), bn["bias"].to(DEV).requires_grad_(True),
                                     bn["running_mean"].to(DEV), bn["running_var"].to(DEV), bn["nbt"].to(DEV), True, 0, 0, False)
        (y.float() * T(synth.normal("dual/cot", tuple(y.shape))).to(DEV)).sum().backward()
        monkeypatch.setattr(unet_ops, "call", real)
        assert ("miseg_conv3x3_dgrad_dual" in names) == dual
        outs.append((x0.grad.clone(), x1.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][0].float().abs().max()) > 0 and float(outs[0][1].float().abs().max()) > 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(48, 1, 256, 256), (3, 1, 50, 46), (2, 3, 20, 36)])
def test_stem_input_is_the_cast_image_padded_with_zero_channels(shape, dtype, monkeypatch):
    """miseg_cast_pad (both kernels: the one-channel 16-byte-vector form, and the generic one) bit for bit.  (With the stem's own kernels
    -- the shipped path for a one-channel image in 16-bit storage -- stem_input only DESCRIBES the operand: same shape and type, the
    fp32 image attached; that form is checked by test_stem_kernels_equal_the_mfma_path_and_the_fp32_reference.)"""
    from miseg_amd import unet_ops
    img = T(synth.normal("stem_input/img", shape)).to(DEV)
    if shape[1] == 1 and dtype != torch.float32:
        desc = unet_ops.stem_input(img, dtype)
        assert desc.dtype == dtype and tuple(desc.shape) == (shape[0], unet_ops.vec_of(dtype), shape[2], shape[3]) and desc._miseg_stem_f32 is img
    monkeypatch.setattr(unet_ops, "_STEM_KERNELS", False)
    out = unet_ops.stem_input(img, dtype)
    vec = unet_ops.vec_of(dtype)
    cp = (shape[1] + vec - 1) // vec * vec
    assert tuple(out.shape) == (shape[0], cp, shape[2], shape[3]) and out.dtype == dtype and out.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out[:, :shape[1]], img.to(dtype)) and float(out[:, shape[1]:].float().abs().max()) == 0.0
