"""GPU parity: local / global IIC mutual information and the cluster heads (HIP, through the C ABI)
against the CPU oracle and the golden vectors generated from the reference.

Tolerance policy (stated once): the MI loss is a small difference of O(1) entropies, so two correct
fp32 evaluations with different summation order differ by ~1e-7 absolute -- not 1e-5 relative to a
loss that can itself be 1e-6.  We therefore require
  * the raw displacement joint (the contraction): 1e-5 relative to its largest entry;
  * the loss: |hip - truth| <= 1e-5 * (sum of |summands|) where truth is the fp64 oracle, i.e. 1e-5
    relative to the entropy scale, AND no worse than 4x the reference's own fp32 deviation + 1e-7;
  * gradients: 1e-4 of the gradient scale vs the fp64 oracle (fp32 reference itself is at that level).
"""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import heads as OH
from oracle import iic as OI

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda"

LOCAL_CASES = [(3, 5, 12, 10, 2), (4, 20, 32, 32, 1), (4, 20, 32, 32, 3), (2, 8, 64, 64, 3)]


def ops():
    from miseg_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("n,k,h,w,p", LOCAL_CASES)
def test_local_joint_and_loss_vs_golden(golden, n, k, h, w, p):
    g = golden("iic")
    key = f"local_f32_n{n}_k{k}_h{h}_w{w}_p{p}"
    x = T(synth.probs(key + "/x", (n, k, h, w))).to(DEV).requires_grad_(True)
    y = T(synth.probs(key + "/y", (n, k, h, w))).to(DEV).requires_grad_(True)
    win = [(0, h, 0, w)]
    raw = ops().local_mi_raw_joint(x.detach(), y.detach(), p, win)[0].cpu().numpy()        # [T,T,K,K]
    ref_raw = np.transpose(g[f"{key}/raw_kktt"], (2, 3, 0, 1))
    np.testing.assert_allclose(raw, ref_raw, rtol=1e-5, atol=1e-5 * np.abs(ref_raw).max())
    loss = ops().local_mi_losses(x, y, p, win)[0]
    # fp64 truth on the same fp32 inputs
    x64, y64 = x.detach().cpu().double().requires_grad_(True), y.detach().cpu().double().requires_grad_(True)
    truth = OI.iid_seg_loss(x64, y64, p)
    gx64, gy64 = torch.autograd.grad(truth, [x64, y64])
    ref_dev = abs(float(g[f"{key}/loss"]) - float(truth))
    _record(key, loss, truth)
    _record(key + "[reference fp32]", g[f"{key}/loss"], truth)
    assert abs(float(loss) - float(truth)) <= 4 * ref_dev + 1e-7, (float(loss), float(truth), ref_dev)
    # north_star's figure taken literally, although these losses are 4e-3 .. 8e-3 = differences of O(1) entropies: measured 1.4e-6 ..
    # 6.2e-6 relative (gpurun_out/mi_rel_errors.json; the reference's own fp32 evaluation is at 3e-6 .. 3.7e-5 on the same inputs)
    assert abs(float(loss) - float(truth)) <= 1e-5 * abs(float(truth)), (float(loss), float(truth))
    loss.backward()
    gscale = float(gx64.abs().max())
    np.testing.assert_allclose(x.grad.cpu().numpy(), gx64.numpy(), rtol=0, atol=1e-4 * gscale + 1e-12)
    np.testing.assert_allclose(y.grad.cpu().numpy(), gy64.numpy(), rtol=0, atol=1e-4 * gscale + 1e-12)


@pytest.mark.parametrize("n,k,h,w,p,patch,use_mask", [(2, 4, 100, 100, 1, 32, False), (2, 4, 100, 100, 1, 32, True),
                                                       (2, 6, 64, 64, 2, 1024, False), (2, 5, 48, 40, 1, 16, True),
                                                       (1, 3, 512, 512, 3, 128, False)])
def test_patch_local_mi(golden, n, k, h, w, p, patch, use_mask):
    """Overlapping patch windows (incl. the clamped last window) + optional mask, one batched launch.  The last case is the
    BASELINE configs[3] geometry (512x512 map, 7x7 overlapping 128x128 patches, displacement +-3; ref iic_loss.py:152-189)."""
    g = golden("iic")
    key = f"patch_f64_n{n}_k{k}_h{h}_w{w}_p{p}_ps{patch}_m{int(use_mask)}"
    x64 = T(synth.probs(key + "/x", (n, k, h, w), np.float64)).float().double().requires_grad_(True)
    y64 = T(synth.probs(key + "/y", (n, k, h, w), np.float64)).float().double().requires_grad_(True)
    m = T(synth.mask(key + "/mask", (n, 1, h, w))) if use_mask else None
    truth = OI.iid_seg_small_patch_loss(x64, y64, p, patch, mask=None if m is None else m.double())
    gx64, gy64 = torch.autograd.grad(truth, [x64, y64])
    x = x64.detach().float().to(DEV).requires_grad_(True)
    y = y64.detach().float().to(DEV).requires_grad_(True)
    wins = OI.patch_windows(h, w, (patch, patch), (patch // 2, patch // 2))
    losses = ops().local_mi_losses(x, y, p, wins, mask=None if m is None else m.to(DEV))
    loss = losses.sum() / float(len(wins))
    _record(key, loss, truth)
    assert abs(float(loss) - float(truth)) <= 1e-5 * abs(float(truth)), (float(loss), float(truth))      # measured 1.4e-7 .. 2.6e-6
    assert abs(float(truth) - float(g[f"{key}/loss"])) <= 1e-9 * max(1.0, abs(float(truth)))   # the oracle IS the reference here (fp64 golden)
    loss.backward()
    gscale = float(gx64.abs().max())
    # ... and the gradient against the reference's own fp64 autograd result (fingerprint), not only the oracle's
    synth.check_fingerprint(x.grad.cpu().numpy(), synth.fp_unpack(g, f"{key}/gx"), f"{key}/gx", rtol=0, atol=2e-3 * gscale + 1e-12)
    synth.check_fingerprint(y.grad.cpu().numpy(), synth.fp_unpack(g, f"{key}/gy"), f"{key}/gy", rtol=0, atol=2e-3 * gscale + 1e-12)
    np.testing.assert_allclose(x.grad.cpu().numpy(), gx64.numpy(), rtol=0, atol=2e-3 * gscale + 1e-12)
    np.testing.assert_allclose(y.grad.cpu().numpy(), gy64.numpy(), rtol=0, atol=2e-3 * gscale + 1e-12)


_REL_ERRORS = {}


def _record(name, got, truth):
    """Achieved relative error of every MI loss this file evaluates -> gpurun_out/mi_rel_errors.json (quoted in DESIGN.md)."""
    import json
    import os
    _REL_ERRORS[name] = {"hip": float(got), "truth_fp64": float(truth), "rel": abs(float(got) - float(truth)) / (abs(float(truth)) + 1e-300)}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "mi_rel_errors.json"), "w") as f:
        json.dump(_REL_ERRORS, f, indent=1, sort_keys=True)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16f8"])
def test_peaked_inputs_hold_1e5_relative(golden, precision):
    """north_star: 'softmax/MI loss within 1e-5 rel fp32'.  On correlated, peaked inputs (MI 0.18 .. 1.45; golden cases gpeak_* /
    lpeak_* / ppeak_* produced by the reference) the bound is applied LITERALLY: |hip - reference| <= 1e-5 |reference| against the
    reference's own fp32 result, and against an fp64 evaluation of the same fp32 inputs; gradients 1e-4 of their scale.
    bf16x3 / f16f8 = the matrix-core arithmetics of the local-MI kernels (hi/lo-split bf16 MFMA; f16 hi x hi + fp8 cross terms in the
    backward, the bench's arithmetic), held to the same bound."""
    g = golden("iic")
    ops().set_mi_precision(precision)
    try:
        for n, k in ((16, 20), (64, 5)):
            key = f"gpeak_f32_n{n}_k{k}"
            xs, ys = synth.peaked_pair(key, (n, k))
            x, y = T(xs).to(DEV).unsqueeze(0).requires_grad_(True), T(ys).to(DEV).unsqueeze(0).requires_grad_(True)
            loss, loss_nl, joint = ops().global_mi(x, y)
            truth = OI.iid_loss(T(xs).double(), T(ys).double())[0]
            _record(f"{key}[{precision}]", loss[0], truth)
            for ref in (float(g[f"{key}/loss"]), float(truth)):
                assert abs(float(loss[0]) - ref) <= 1e-5 * abs(ref), (key, float(loss[0]), ref)
            assert abs(float(loss_nl[0]) - float(g[f"{key}/loss_no_lamb"])) <= 1e-5 * abs(float(g[f"{key}/loss_no_lamb"]))
            np.testing.assert_allclose(joint[0].detach().cpu().numpy(), g[f"{key}/joint"], rtol=1e-5, atol=1e-8)
            loss[0].backward()
            sc = float(np.abs(g[f"{key}/gx"]).max())
            np.testing.assert_allclose(x.grad[0].cpu().numpy(), g[f"{key}/gx"], rtol=1e-4, atol=1e-4 * sc)
            np.testing.assert_allclose(y.grad[0].cpu().numpy(), g[f"{key}/gy"], rtol=1e-4, atol=1e-4 * sc)
        cases = [(f"lpeak_f32_n{n}_k{k}_h{h}_w{w}_p{p}", (n, k, h, w), p, None) for n, k, h, w, p in LOCAL_CASES] + \
                [(f"ppeak_f32_n{n}_k{k}_h{h}_w{w}_p{p}_ps{ps}", (n, k, h, w), p, ps) for n, k, h, w, p, ps in ((2, 4, 100, 100, 1, 32), (1, 20, 96, 96, 3, 32))]
        for key, shape, p, patch in cases:
            xs, ys = synth.peaked_pair(key, shape)
            x, y = T(xs).to(DEV).requires_grad_(True), T(ys).to(DEV).requires_grad_(True)
            h, w = shape[2:]
            wins = [(0, h, 0, w)] if patch is None else OI.patch_windows(h, w, (patch, patch), (patch // 2, patch // 2))
            loss = ops().local_mi_losses(x, y, p, wins).sum() / float(len(wins))
            x64, y64 = T(xs).double().requires_grad_(True), T(ys).double().requires_grad_(True)
            truth = OI.iid_seg_loss(x64, y64, p) if patch is None else OI.iid_seg_small_patch_loss(x64, y64, p, patch)
            gx64, gy64 = torch.autograd.grad(truth, [x64, y64])
            _record(f"{key}[{precision}]", loss, truth)
            for ref in (float(g[f"{key}/loss"]), float(truth)):
                assert abs(float(loss) - ref) <= 1e-5 * abs(ref), (key, float(loss), ref)
            loss.backward()
            sc = float(gx64.abs().max())
            np.testing.assert_allclose(x.grad.cpu().numpy(), gx64.numpy(), rtol=0, atol=1e-4 * sc)
            np.testing.assert_allclose(y.grad.cpu().numpy(), gy64.numpy(), rtol=0, atol=1e-4 * sc)
            synth.check_fingerprint(x.grad.cpu().numpy(), synth.fp_unpack(g, f"{key}/gx"), f"{key}/gx", rtol=1e-4, atol=1e-4 * sc)
    finally:
        ops().set_mi_precision("fp32")


def test_local_mi_full_size_properties():
    """BASELINE cfg2 shape (16x20x256x256, pad 3): size-independent properties instead of a CPU oracle run:
    (1) sum_{i,j} R[a,b,i,j] == number of valid (pixel, displacement) pairs * N  (both inputs are simplexes);
    (2) swapping x<->y transposes (i,j) and mirrors the displacement; (3) joint is linear in x."""
    n, k, h, w, p = 16, 20, 256, 256, 3
    gen = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(n, k, h, w, generator=gen).softmax(1).to(DEV)
    y = torch.randn(n, k, h, w, generator=gen).softmax(1).to(DEV)
    win = [(0, h, 0, w)]
    raw = ops().local_mi_raw_joint(x, y, p, win)[0]
    t = 2 * p + 1
    counts = torch.empty(t, t)
    for a in range(t):
        for b in range(t):
            counts[a, b] = (h - abs(a - p)) * (w - abs(b - p)) * n
    np.testing.assert_allclose(raw.sum(dim=(2, 3)).cpu().numpy(), counts.numpy(), rtol=2e-5)
    raw_sw = ops().local_mi_raw_joint(y, x, p, win)[0]
    np.testing.assert_allclose(raw_sw.cpu().numpy(), raw.flip(0, 1).transpose(2, 3).cpu().numpy(), rtol=1e-4, atol=1e-3)
    raw2 = ops().local_mi_raw_joint(0.5 * x, y, p, win)[0]
    np.testing.assert_allclose(raw2.cpu().numpy(), 0.5 * raw.cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("n,k", [(7, 5), (16, 20)])
def test_global_mi(golden, n, k):
    g = golden("iic")
    key = f"global_f32_n{n}_k{k}"
    x = T(synth.probs(key + "/x", (n, k))).to(DEV)
    y = T(synth.probs(key + "/y", (n, k))).to(DEV)
    xs = torch.stack([x, y, x]).requires_grad_(True)    # three "sub-heads" in one launch
    ys = torch.stack([y, x, x]).requires_grad_(True)
    loss, loss_nl, joint = ops().global_mi(xs, ys)
    np.testing.assert_allclose(joint[0].detach().cpu().numpy(), g[f"{key}/joint"], rtol=1e-5, atol=1e-8)
    x64, y64 = x.cpu().double().requires_grad_(True), y.cpu().double().requires_grad_(True)
    truth, truth_nl, _ = OI.iid_loss(x64, y64)
    ref_dev = abs(float(g[f"{key}/loss"]) - float(truth))
    _record(key, loss[0], truth)
    assert abs(float(loss[0]) - float(truth)) <= 1e-5 * abs(float(truth))       # measured 2.8e-6 / 3.5e-6
    assert abs(float(loss[0]) - float(truth)) <= 4 * ref_dev + 2e-7
    assert abs(float(loss_nl[0]) - float(truth_nl)) <= 4 * ref_dev + 2e-7
    gx64, gy64 = torch.autograd.grad(truth, [x64, y64])
    loss[0].backward()
    gscale = float(gx64.abs().max())
    np.testing.assert_allclose(xs.grad[0].cpu().numpy(), gx64.numpy(), rtol=0, atol=2e-4 * gscale + 1e-9)
    np.testing.assert_allclose(ys.grad[0].cpu().numpy(), gy64.numpy(), rtol=0, atol=2e-4 * gscale + 1e-9)
    assert float(xs.grad[1].abs().max()) == 0.0   # untouched sub-heads get exactly zero


@pytest.mark.parametrize("n,k", [(7, 5), (16, 20)])
def test_global_mi_pair_is_bitwise_the_two_tensor_form(n, k):
    """`global_mi_pair(prob[S, 2N, K])` (the epocher's layout: the two views stacked along dim 1, read in place, ONE gradient
    tensor) against `global_mi(prob[:, :N], prob[:, N:])` (pinned to the golden vectors above): same kernels, same order of
    operations -> identical bits for losses, joints and gradients."""
    prob = T(synth.probs(f"gpair_n{n}_k{k}", (3 * 2 * n, k))).view(3, 2 * n, k).to(DEV)
    wts = torch.tensor([1.0, -2.0, 0.5], device=DEV)
    a = prob.clone().requires_grad_(True)
    la, la_nl, ja = ops().global_mi_pair(a)
    (la * wts).sum().backward()
    b = prob.clone().requires_grad_(True)
    lb, lb_nl, jb = ops().global_mi(b[:, :n], b[:, n:])
    (lb * wts).sum().backward()
    assert torch.equal(la, lb) and torch.equal(la_nl, lb_nl) and torch.equal(ja, jb)
    assert torch.equal(a.grad, b.grad)
    with pytest.raises(Exception):
        ops().global_mi_pair(prob[:, :2 * n - 1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_local_head_fwd_bwd(golden, dtype):
    """LocalClusterHead (linear) incl. the fused sample gather + flip replay, vs oracle autograd."""
    g = golden("heads")
    sd = OH.init_local_cluster_head(8, 6, 3, "linear", seed=6)
    feat = T(synth.normal("dec_linear_norm0/feat", (3, 8, 10, 12)))
    w = torch.stack([sd[f"_headers.{s}.0.weight"].view(6, 8) for s in range(3)])
    b = torch.stack([sd[f"_headers.{s}.0.bias"] for s in range(3)])
    src = torch.tensor([0, 1, 2], dtype=torch.int32, device=DEV)
    fd = feat.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    prob = ops().local_head(fd, wd, bd, src, None)
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-3)
    if dtype == torch.float32:
        for s in range(3):
            np.testing.assert_allclose(prob[s].detach().cpu().numpy(), g[f"dec_linear_norm0/out{s}"], **tol)
    # gather + flips + backward against the oracle on the (dtype-rounded) feature
    f_ref = fd.detach().float().cpu().contiguous().requires_grad_(True)
    sd_ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    order, dec = [2, 0, 1], [[True, False], [False, True], [True, True]]   # sources must be distinct (gfeat scatter)
    gathered = torch.stack([f_ref[i] for i in order])
    from oracle import losses as OL
    ref_probs = OH.local_cluster_head(sd_ref, OL.apply_flips(gathered, dec))
    cot = T(synth.normal("head/cot", (3, 3, 6, 10, 12)))
    sum((p * c).sum() for p, c in zip(ref_probs, cot)).backward()
    src2 = torch.tensor(order, dtype=torch.int32, device=DEV)
    flips = ops().flips_to_tensor(dec, DEV)
    prob2 = ops().local_head(fd, wd, bd, src2, flips)
    for s in range(3):
        np.testing.assert_allclose(prob2[s].detach().cpu().numpy(), ref_probs[s].detach().numpy(), **tol)
    (prob2 * cot.to(DEV)).sum().backward()
    gw_ref = torch.stack([sd_ref[f"_headers.{s}.0.weight"].grad.view(6, 8) for s in range(3)])
    gb_ref = torch.stack([sd_ref[f"_headers.{s}.0.bias"].grad for s in range(3)])
    gt = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    np.testing.assert_allclose(wd.grad.cpu().numpy(), gw_ref.numpy(), **gt)
    np.testing.assert_allclose(bd.grad.cpu().numpy(), gb_ref.numpy(), **gt)
    np.testing.assert_allclose(fd.grad.float().cpu().numpy(), f_ref.grad.numpy(), **gt)


def test_global_head_fwd_bwd(golden):
    g = golden("heads")
    sd = OH.init_cluster_head(32, 6, 3, "linear", seed=5)
    feat = T(synth.normal("enc_linear_norm0/feat", (5, 32, 6, 6)))
    w = torch.stack([sd[f"_headers.{s}.2.weight"] for s in range(3)])
    b = torch.stack([sd[f"_headers.{s}.2.bias"] for s in range(3)])
    src = torch.arange(5, dtype=torch.int32, device=DEV)
    fd = feat.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    prob = ops().global_head(fd, wd, bd, src)
    for s in range(3):
        np.testing.assert_allclose(prob[s].detach().cpu().numpy(), g[f"enc_linear_norm0/out{s}"], rtol=1e-5, atol=1e-7)
    f_ref = feat.clone().requires_grad_(True)
    sd_ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = OH.cluster_head(sd_ref, f_ref)
    cot = T(synth.normal("ghead/cot", (3, 5, 6)))
    sum((p * c).sum() for p, c in zip(ref, cot)).backward()
    (prob * cot.to(DEV)).sum().backward()
    np.testing.assert_allclose(wd.grad.cpu().numpy(), torch.stack([sd_ref[f"_headers.{s}.2.weight"].grad for s in range(3)]).numpy(),
                               rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(bd.grad.cpu().numpy(), torch.stack([sd_ref[f"_headers.{s}.2.bias"].grad for s in range(3)]).numpy(),
                               rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(fd.grad.cpu().numpy(), f_ref.grad.numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("n,k,h,w,p", [(4, 20, 32, 32, 3), (4, 20, 32, 32, 1), (3, 20, 70, 90, 3), (16, 20, 256, 256, 3),
                                       (2, 20, 37, 45, 3), (1, 20, 7, 5, 1), (5, 20, 9, 131, 3)])     # odd heights, maps smaller than a tile, 2+ strips with a ragged last one
def test_joint_bf16_split_matches_fp32_kernel(n, k, h, w, p):
    """bf16-MFMA joint with hi/lo operand split (3 products) vs the exact fp32-MFMA kernel and, for the small shapes,
    the fp64 oracle; plain bf16 (1 product) at bf16 tolerance.  Includes a non-multiple-of-tile shape and a window crop."""
    gen = torch.Generator(device="cpu").manual_seed(n * 1000 + h)
    x = torch.randn(n, k, h, w, generator=gen).softmax(1).to(DEV)
    y = torch.randn(n, k, h, w, generator=gen).softmax(1).to(DEV)
    for win in ([(0, h, 0, w)], [(3, h - 5, 2, w - 7)] if 12 <= h < 100 else [(0, h, 0, w)]):
        ref = ops().local_mi_raw_joint(x, y, p, win, precision="fp32")
        scale = float(ref.abs().max())
        split = ops().local_mi_raw_joint(x, y, p, win, precision="bf16x3")
        # every product carries <= 2^-16 relative error (the lo x lo term is dropped) and all products are positive: the errors average
        # out over the pixels of a window, so the bound tightens with their number (2e-6 from ~230 pixels on; a 7 x 5 map gets 5e-6)
        npix = n * (win[0][1] - win[0][0]) * (win[0][3] - win[0][2])
        assert float((split - ref).abs().max()) <= max(2e-6, 3e-5 / npix ** 0.5) * scale, float((split - ref).abs().max()) / scale
        plain = ops().local_mi_raw_joint(x, y, p, win, precision="bf16")
        assert float((plain - ref).abs().max()) <= 2e-3 * scale, float((plain - ref).abs().max()) / scale
        if h <= 32:
            truth = OI.local_joint_raw(x[:, :, win[0][0]:win[0][1], win[0][2]:win[0][3]].double().cpu(),
                                       y[:, :, win[0][0]:win[0][1], win[0][2]:win[0][3]].double().cpu(), p)
            e_split = float((split[0].double().cpu() - truth).abs().max())
            e_ref = float((ref[0].double().cpu() - truth).abs().max())
            assert e_split <= 4 * e_ref + max(1e-6, 3e-5 / npix ** 0.5) * scale, (e_split, e_ref)


@pytest.mark.parametrize("precision", ["bf16x3", "f16f8"])
@pytest.mark.parametrize("n,h,w,p", [(4, 32, 32, 3), (4, 32, 32, 1), (3, 70, 90, 3), (2, 37, 45, 3), (1, 7, 5, 1), (5, 9, 131, 3)])
def test_local_mi_bf16_split_fwd_bwd_matches_fp64(n, h, w, p, precision):
    """Whole local-MI op (joint -> loss -> backward) in 'bf16x3' / 'f16f8' precision vs the fp64 oracle: same bounds as the fp32 path."""
    k = 20
    gen = torch.Generator(device="cpu").manual_seed(n * 100 + h + p)
    x0 = torch.randn(n, k, h, w, generator=gen).softmax(1)
    y0 = torch.randn(n, k, h, w, generator=gen).softmax(1)
    x64, y64 = x0.double().requires_grad_(True), y0.double().requires_grad_(True)
    truth = OI.iid_seg_loss(x64, y64, p)
    gx64, gy64 = torch.autograd.grad(truth, [x64, y64])
    ops().set_mi_precision(precision)
    try:
        x, y = x0.to(DEV).requires_grad_(True), y0.to(DEV).requires_grad_(True)
        loss = ops().local_mi_losses(x, y, p, [(0, h, 0, w)])[0]
        loss.backward()
    finally:
        ops().set_mi_precision("fp32")
    assert abs(float(loss) - float(truth)) <= 5e-7, (float(loss), float(truth))
    gscale = float(gx64.abs().max())
    np.testing.assert_allclose(x.grad.cpu().numpy(), gx64.numpy(), rtol=0, atol=2e-4 * gscale + 1e-12)
    np.testing.assert_allclose(y.grad.cpu().numpy(), gy64.numpy(), rtol=0, atol=2e-4 * gscale + 1e-12)


@pytest.mark.parametrize("s,ub,k,h,w,p,patch", [(3, 2, 6, 24, 20, 1, 1024), (2, 2, 5, 40, 40, 2, 16)])
def test_local_mi_heads_equals_per_head_calls(s, ub, k, h, w, p, patch):
    """The one-node multi-head path (gradient written into one [S,2UB,K,H,W] buffer) is the per-sub-head loop of
    ref semi_seg/epocher.py:264-272, bit for bit (same kernels, same operands)."""
    from contrastyou.losses.iic_loss import IIDSegmentationSmallPathLoss
    crit = IIDSegmentationSmallPathLoss(padding=p, patch_size=patch)
    base = T(synth.probs(f"heads_s{s}_ub{ub}_k{k}_h{h}_w{w}", (s * 2 * ub, k, h, w))).view(s, 2 * ub, k, h, w).to(DEV)
    a = base.clone().requires_grad_(True)
    fused = crit.forward_heads(a, ub)
    (fused * torch.arange(1, s + 1, device=DEV)).sum().backward()
    b = base.clone().requires_grad_(True)
    loop = torch.stack([crit(q[:ub], q[ub:]) for q in b])
    (loop * torch.arange(1, s + 1, device=DEV)).sum().backward()
    np.testing.assert_allclose(fused.detach().cpu().numpy(), loop.detach().cpu().numpy(), rtol=1e-6, atol=1e-8)
    np.testing.assert_array_equal(a.grad.cpu().numpy(), b.grad.cpu().numpy())


def test_simplex_checks_immediate_and_deferred():
    """ref iic_loss.py:28-29 asserts simplex inline; here the same AssertionError is raised inline outside a
    ``deferred`` block and at the block owner's fetch inside one (semi_seg/epocher.py _Pending)."""
    from contrastyou.losses.iic_loss import IIDSegmentationLoss, simplex
    from miseg_amd import checks
    from semi_seg.epocher import _Pending
    good = T(synth.probs("simplex/good", (2, 5, 9, 7))).to(DEV)
    bad = good.clone()
    bad[1, :, 4, 3] *= 1.01
    assert simplex(good) and not simplex(bad)
    assert int(checks.simplex_violations(bad, 1)) == 1
    assert int(checks.simplex_violations(good.permute(0, 2, 3, 1).contiguous(), 3)) == 0
    nan = good.clone()
    nan[0, 2, 0, 0] = float("nan")
    assert int(checks.simplex_violations(nan, 1)) == 1
    crit = IIDSegmentationLoss(padding=1)
    with pytest.raises(AssertionError):
        crit(bad.requires_grad_(True), good.clone().requires_grad_(True))
    pend = _Pending()
    with checks.deferred(pend.checks):
        loss = crit(bad, good.clone().requires_grad_(True))   # no exception yet
        pend.put("loss", loss)
    with pytest.raises(AssertionError):
        pend.fetch()
    assert pend.fetch() == {}   # flags are consumed


def test_local_head_counts_simplex_violations_for_free():
    """ops.local_head's kernel evaluates the consumer's ``assert simplex`` (ref iic_loss.py:28-29) while the
    probabilities are in registers; checks.simplex_violations must hand back that counter for the untouched tensor,
    and it must agree with the stand-alone check."""
    from miseg_amd import checks
    feat = torch.randn(4, 16, 12, 20, device=DEV).to(memory_format=torch.channels_last)
    w, b = torch.randn(3, 6, 16, device=DEV), torch.randn(3, 6, device=DEV)
    src = torch.arange(4, dtype=torch.int32, device=DEV)
    prob = ops().local_head(feat, w, b, src, None, 1.0)
    fused = checks.simplex_violations(prob, 2)
    assert fused is prob._miseg_simplex[1] and int(fused) == 0
    del prob._miseg_simplex
    assert int(checks.simplex_violations(prob, 2)) == 0
    b_bad = b.clone()
    b_bad[1, 2] = float("nan")
    bad = ops().local_head(feat, w, b_bad, src, None, 1.0)
    n_fused = int(checks.simplex_violations(bad, 2))
    del bad._miseg_simplex
    assert n_fused == int(checks.simplex_violations(bad, 2)) == 4 * 12 * 20   # every pixel of sub-head 1
    bad2 = ops().local_head(feat, w, b, src, None, 1.0)
    bad2[0, 0, 0, 0, 0] += 0.5          # in-place edit bumps the version: the cached counter no longer applies
    assert int(checks.simplex_violations(bad2, 2)) == 1


@pytest.mark.parametrize("precision", ["bf16x3", "f16f8"])
@pytest.mark.parametrize("pad,h,w,patch,s,ub", [(3, 64, 64, 32, 3, 2), (1, 48, 80, 32, 3, 2), (3, 512, 512, 128, 2, 1)])
def test_bf16x3_batched_heads_with_overlapping_patches(pad, h, w, patch, s, ub, precision):
    """K=20 (the shipped cluster count) puts `forward_heads` on the batched bf16 kernels: S sub-heads x P overlapping
    patch windows in one launch forward, one launch per colour group backward (accumulating).  Compared with the exact-fp32
    kernels run one sub-head at a time (themselves pinned to the golden vectors above): loss to 5e-7 absolute; gradients
    to 2e-4 of the gradient scale -- each side is held to 1e-4 against the fp64 oracle elsewhere in this file, and here
    two fp32-class evaluations are compared with each other."""
    from contrastyou.losses.iic_loss import IIDSegmentationSmallPathLoss
    k = 20     # last case: BASELINE configs[3] geometry at the shipped cluster count -- 49 overlapping 128x128 windows, +-3
    crit = IIDSegmentationSmallPathLoss(padding=pad, patch_size=patch)
    base = T(synth.probs(f"bf16heads_p{pad}_h{h}_w{w}", (s * 2 * ub, k, h, w))).view(s, 2 * ub, k, h, w).to(DEV)
    wts = torch.arange(1, s + 1, device=DEV, dtype=torch.float32)
    try:
        ops().set_mi_precision(precision)
        a = base.clone().requires_grad_(True)
        fused = crit.forward_heads(a, ub)
        (fused * wts).sum().backward()
    finally:
        ops().set_mi_precision("fp32")
    b = base.clone().requires_grad_(True)
    loop = torch.stack([crit(q[:ub], q[ub:]) for q in b])
    (loop * wts).sum().backward()
    np.testing.assert_allclose(fused.detach().cpu().numpy(), loop.detach().cpu().numpy(), rtol=0, atol=5e-7)
    gscale = float(b.grad.abs().max())
    np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=0, atol=2e-4 * gscale)
    if h == 512:   # and individual windows of sub-head 0 against the fp64 oracle (ref iic_loss.py:107-149 per window): first, an
        # interior one, the last (clamped) one -- the whole 49-window fp64 sum costs minutes of CPU, the kernels are the same per window
        from contrastyou.losses.iic_loss import _windows
        wins = _windows(h, w, (patch, patch), (patch // 2, patch // 2))
        assert len(wins) == 49 and wins[-1] == (384, 512, 384, 512)
        try:
            ops().set_mi_precision(precision)
            per_window = ops().local_mi_heads(base.clone().requires_grad_(True), ub, pad, wins).detach().cpu()   # [S, 49]
        finally:
            ops().set_mi_precision("fp32")
        np.testing.assert_allclose(per_window.mean(1).numpy(), fused.detach().cpu().numpy(), rtol=0, atol=2e-7)
        for wi in (0, 24, 48):
            h0, h1, w0, w1 = wins[wi]
            truth = OI.iid_seg_loss(base[0, :ub, :, h0:h1, w0:w1].double().cpu(), base[0, ub:, :, h0:h1, w0:w1].double().cpu(), pad)
            assert abs(float(per_window[0, wi]) - float(truth)) <= 5e-7, (wi, float(per_window[0, wi]), float(truth))


@pytest.mark.parametrize("h,w", [(64, 64), (37, 45)])
def test_head_backward_joins_an_offered_gradient(h, w):
    """ops._GradJoin: when another consumer of the tapped feature (DeConv_1x1) has offered the input gradient it wrote, the head's
    backward adds its own into that tensor in the kernel epilogue (`miseg_head_local_bwd_acc`) and returns none -- autograd then has
    nothing to add.  Result = offered + the head's stand-alone gradient up to ONE bf16 rounding of the sum (the two-kernel form rounds
    twice); rows outside the gathered range keep the offered values bit for bit; weight gradients are unchanged."""
    from miseg_amd import ops as O
    bsz, c, s, k = 6, 16, 5, 20
    feat = T(synth.normal(f"join/{h}/f", (bsz, c, h, w))).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = T(synth.normal(f"join/{h}/w", (s, k, c), scale=0.3)).to(DEV)
    bs = T(synth.normal(f"join/{h}/b", (s, k), scale=0.1)).to(DEV)
    src = O.arange_i32(2, 6, DEV)
    flips = torch.tensor([0, 1, 2, 3], dtype=torch.int32, device=DEV)
    cot = T(synth.normal(f"join/{h}/cot", (s, 4, k, h, w))).to(DEV)
    base = T(synth.normal(f"join/{h}/g0", (bsz, c, h, w))).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    O._GradJoin.enabled = True      # opt-in feature (MISEG_GRAD_JOIN=1)

    def run(offer):
        f = feat.clone(memory_format=torch.preserve_format).requires_grad_(True)
        w_, b_ = wt.clone().requires_grad_(True), bs.clone().requires_grad_(True)
        prob = O.local_head(f, w_, b_, src, flips, 1.0)
        O._GradJoin.clear()
        g0 = base.clone(memory_format=torch.preserve_format)
        if offer:      # what a consumer that ran earlier in the same backward pass leaves behind (offer() itself only works inside one)
            ev = torch.cuda.Event()
            ev.record()
            O._GradJoin._offers[(f.data_ptr(), tuple(f.shape))] = (g0, ev, torch.cuda.current_stream())
        (prob * cot).sum().backward()
        return f.grad, g0, w_.grad, b_.grad

    g_alone, _, gw_a, gb_a = run(False)
    g_none, g_joined, gw_j, gb_j = run(True)
    assert g_none is None and not O._GradJoin._offers
    ref = base.float() + g_alone.float()
    assert torch.equal(g_joined[:2], base[:2])                                        # rows that are not gathered: untouched
    err = (g_joined.float() - ref).abs()
    bound = 2.0 ** -8 * (base.float().abs() + 2 * g_alone.float().abs()) + 1e-6        # rounding of the sum + the bf16 rounding of g_alone itself
    assert bool((err <= bound).all()), float((err / bound).max())
    assert torch.equal(gw_a, gw_j) and torch.equal(gb_a, gb_j)
    O._GradJoin.enabled = os.environ.get("MISEG_GRAD_JOIN", "0") == "1"


@pytest.mark.parametrize("c,h,w", [(32, 40, 48), (16, 6, 10), (16, 72, 36)])
def test_local_head_forward_mfma_vs_float64(c, h, w):
    """The shipped taps (bf16 features, S=5 x K=20, C in {16, 32}) compute the logits on the matrix pipe with the fp32
    weights split into three bf16 planes (heads.hip: head_local_fwd_mfma_kernel).  Against a float64 evaluation of
    softmax((W f + b) / T) on the gathered / flipped features (ref contrastyou/trainer/_utils.py:137-168,
    semi_seg/epocher.py:258-273): 1e-5 relative on the probabilities, the same bound as the fp32 register kernel.
    Shapes: a ragged tail (HW % 1024 != 0), W % 4 != 0, and several blocks per sample."""
    torch.manual_seed(11)
    feat = torch.randn(5, c, h, w, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(5, 20, c, device=DEV) * 0.7
    b = torch.randn(5, 20, device=DEV)
    src = torch.tensor([4, 0, 2, 1], dtype=torch.int32, device=DEV)
    flips = torch.tensor([3, 0, 1, 2], dtype=torch.int32, device=DEV)
    temp = 0.8
    prob = ops().local_head(feat, wt, b, src, flips, temp)
    from miseg_amd import checks
    assert int(checks.simplex_violations(prob, 2)) == 0
    g = feat.double()[src.long()]
    g = torch.stack([gi.flip([d for d, on in ((1, f & 1), (2, f & 2)) if on]) if f else gi for gi, f in zip(g, flips.tolist())])
    z = torch.einsum("skc,mchw->smkhw", wt.double(), g) + b.double()[:, None, :, None, None]
    ref = torch.softmax(z / temp, dim=2)
    assert prob.shape == ref.shape
    assert float(((prob.double() - ref).abs() / (ref + 1e-9)).max()) <= 1e-5
    assert float((prob.double() - ref).abs().max()) <= 2e-6


@pytest.mark.parametrize("c,h", [(32, 40), (16, 24)])
def test_local_head_backward_shipped_shape_bf16_vs_fp32_kernels(c, h):
    """S=5 x K=20 heads on bf16 features (the shipped taps).  C=32 takes the bf16-MFMA backward (dz as hi+lo bf16 planes, W^T
    hi+lo, exact bf16 features), C=16 the K=20 register path; both against the exact-fp32 kernels on the same (bf16-valued)
    features.  gw / gb: 2e-4 of the gradient scale (hi+lo = 2^-16 per term); gfeat is a bf16 tensor: one bf16 ulp."""
    torch.manual_seed(3)
    b_, m = 6, 4
    feat16 = torch.randn(b_, c, h, h + 8, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(5, 20, c, device=DEV) * 0.3).requires_grad_(True)
    b = torch.randn(5, 20, device=DEV).requires_grad_(True)
    src = torch.tensor([1, 3, 4, 5], dtype=torch.int32, device=DEV)
    flips = torch.tensor([0, 1, 2, 3], dtype=torch.int32, device=DEV)
    outs = []
    for ft in (feat16, feat16.float().contiguous(memory_format=torch.channels_last)):
        f = ft.clone().requires_grad_(True)
        prob = ops().local_head(f, w, b, src, flips, 1.0)
        cot = torch.randn(prob.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
        gf, gw, gb = torch.autograd.grad((prob * cot).sum(), [f, w, b])
        outs.append((prob.detach(), gf.float(), gw, gb))
    (p16, gf16, gw16, gb16), (p32, gf32, gw32, gb32) = outs
    assert torch.allclose(p16, p32, rtol=1e-5, atol=1e-7)
    for a16, a32 in ((gw16, gw32), (gb16, gb32)):
        assert float((a16 - a32).abs().max()) <= 2e-4 * float(a32.abs().max())
    assert float((gf16 - gf32).abs().max()) <= 2.0 ** -7 * float(gf32.abs().max())
    assert float(gf32[0].abs().max()) == 0.0 and float(gf16[0].abs().max()) == 0.0      # untouched source rows stay zero


@pytest.mark.parametrize("pad,p", [(3, 5), (2, 1), (1, 3)])
def test_local_loss_per_displacement_blocks_equal_the_single_block_epilogue(pad, p):
    """miseg_iic_local_loss_fwd_ws (one block per window and displacement, the form the IIC chain uses for pad >= 2) against
    miseg_iic_local_loss_fwd (one block per window): both run the same per-displacement routine, so grad_raw must be
    bit-identical; the loss differs only in the order its (2 pad + 1)^2 terms are added (1e-6 relative)."""
    from miseg_amd import _cabi
    k, t = 20, 2 * pad + 1
    torch.manual_seed(5)
    raw = torch.rand(p, t, t, k, k, device=DEV) * 3.0 + 0.01
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for ws_form in (False, True):
        loss, grad = torch.empty(p, device=DEV), torch.empty_like(raw)
        if ws_form:
            nb = _cabi.query("miseg_iic_local_loss_ws_bytes", pad, p)
            assert nb == p * t * t * 4
            ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
            _cabi.call("miseg_iic_local_loss_fwd_ws", st, raw.data_ptr(), k, pad, p, 1.0, loss.data_ptr(), grad.data_ptr(), ws.data_ptr(), nb)
            with pytest.raises(_cabi.MisegError):
                _cabi.call("miseg_iic_local_loss_fwd_ws", st, raw.data_ptr(), k, pad, p, 1.0, loss.data_ptr(), grad.data_ptr(), ws.data_ptr(), nb - 4)
        else:
            _cabi.call("miseg_iic_local_loss_fwd", st, raw.data_ptr(), k, pad, p, 1.0, loss.data_ptr(), grad.data_ptr())
        outs.append((loss, grad))
    assert torch.equal(outs[0][1], outs[1][1])
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("c", [16, 32])
def test_local_head_backward_zero_fills_only_rows_outside_a_known_source_range(c):
    """With src = ops.arange_i32(start, stop) (how the epocher builds it) the backward allocates the tap gradient uninitialised
    and zero-fills only the rows outside [start, stop): the shipped kernels store every element of the source rows.  Same
    gradient, bit for bit, as with an anonymous src tensor (full zero fill); rows outside the range exactly zero."""
    torch.manual_seed(13)
    bsz, h, w_ = 7, 24, 40
    feat = torch.randn(bsz, c, h, w_, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt, b = torch.randn(5, 20, c, device=DEV) * 0.3, torch.randn(5, 20, device=DEV)
    flips = torch.tensor([0, 3, 1, 2], dtype=torch.int32, device=DEV)
    grads = []
    for src in (ops().arange_i32(2, 6, DEV), torch.arange(2, 6, dtype=torch.int32, device=DEV)):
        junk = torch.full((bsz, c, h, w_), float("nan"), device=DEV, dtype=torch.bfloat16)   # poison what the allocator hands out next
        del junk
        f = feat.clone().requires_grad_(True)
        prob = ops().local_head(f, wt, b, src, flips, 1.0)
        cot = torch.randn(prob.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
        grads.append(torch.autograd.grad((prob * cot).sum(), f)[0])
    assert hasattr(ops().arange_i32(2, 6, DEV), "_miseg_range")
    assert torch.equal(grads[0], grads[1])
    assert float(grads[0][:2].float().abs().max()) == 0.0 and float(grads[0][6:].float().abs().max()) == 0.0
    assert float(grads[0][2:6].float().abs().max()) > 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("h,w_", [(24, 40), (22, 36), (64, 64)])
def test_local_head_backward_recomputing_the_probabilities_is_bit_equal(dtype, h, w_):
    """miseg_head_local_bwd_recompute (the top tap's shipped backward: no probabilities read, the kernel redoes the forward's
    W-planes x features MFMA + softmax) against miseg_head_local_bwd_rows on the forward's saved probabilities: the recomputed
    p is the forward's p bit for bit, so gfeat / gw / gb must be IDENTICAL.  Ragged sizes (h*w % 64 != 0), all four flips."""
    from miseg_amd import _cabi
    torch.manual_seed(21)
    bsz, c, s, k, temp = 7, 16, 5, 20, 0.7
    dt = {torch.bfloat16: 1, torch.float16: 2}[dtype]
    assert _cabi.query("miseg_head_local_bwd_recompute_supported", dt, c, s, k) == 1
    assert _cabi.query("miseg_head_local_bwd_recompute_supported", dt, 32, s, k) == 0
    feat = torch.randn(bsz, c, h, w_, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    wt, b = (torch.randn(s, k, c, device=DEV) * 0.3).contiguous(), torch.randn(s, k, device=DEV)
    src = torch.arange(2, 6, dtype=torch.int32, device=DEV)
    flips = torch.tensor([0, 3, 1, 2], dtype=torch.int32, device=DEV)
    m, st = src.numel(), torch.cuda.current_stream().cuda_stream
    prob = torch.empty(s, m, k, h, w_, device=DEV)
    _cabi.call("miseg_head_local_fwd", st, dt, feat.data_ptr(), bsz, h, w_, c, src.data_ptr(), flips.data_ptr(), m, wt.data_ptr(), b.data_ptr(),
               s, k, temp, prob.data_ptr(), 2e-4, 0)
    gprob = torch.randn(prob.shape, device=DEV)
    nb = _cabi.query("miseg_head_local_bwd_ws_bytes", m, h, w_, c, s, k)
    outs = []
    for recompute in (False, True):
        gfeat = torch.full((m, c, h, w_), float("nan"), device=DEV, dtype=dtype).contiguous(memory_format=torch.channels_last)
        gw, gb, ws = torch.empty_like(wt), torch.empty_like(b), torch.empty(nb, dtype=torch.uint8, device=DEV)
        if recompute:
            _cabi.call("miseg_head_local_bwd_recompute", st, dt, feat.data_ptr(), bsz, h, w_, c, src.data_ptr(), flips.data_ptr(), m, wt.data_ptr(),
                       b.data_ptr(), s, k, temp, gprob.data_ptr(), gfeat.data_ptr(), 2, gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), nb)
        else:
            _cabi.call("miseg_head_local_bwd_rows", st, dt, feat.data_ptr(), bsz, h, w_, c, src.data_ptr(), flips.data_ptr(), m, wt.data_ptr(),
                       s, k, temp, prob.data_ptr(), gprob.data_ptr(), gfeat.data_ptr(), 2, gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), nb)
        outs.append((gfeat, gw, gb))
    for a, r in zip(outs[0], outs[1]):
        assert bool(torch.isfinite(r.float()).all()) and float(r.float().abs().max()) > 0.0
        assert torch.equal(a, r)
    with pytest.raises(_cabi.MisegError):      # another tap shape: refused, not silently computed by another kernel
        _cabi.call("miseg_head_local_bwd_recompute", st, dt, feat.data_ptr(), bsz, h, w_ // 2, 32, src.data_ptr(), flips.data_ptr(), m, wt.data_ptr(),
                   b.data_ptr(), s, k, temp, gprob.data_ptr(), gfeat.data_ptr(), 2, gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), nb)


# ------------------------------------------------------------------------------------------ operand planes of the f16 + fp8 backward
_PLANE_CASES = [(2, 2, 24, 40, [(0, 24, 0, 40)]), (2, 3, 70, 150, [(0, 70, 0, 150)]), (1, 2, 64, 160, [(3, 40, 10, 90), (41, 64, 70, 160)]),
                (2, 2, 50, 90, [(5, 45, 13, 77)]), (1, 1, 128, 256, [(0, 128, 0, 256)])]


def _plane_views(buf, maps, h, w):
    """(p16, p8l, p8h) as [maps, h, w, bytes-per-pixel] views of a plane buffer (layout: csrc/mi_local.h, 4 KiB of slack around each)."""
    out, off = [], 4096
    for bpp in (40, 24, 24):
        out.append(buf[off:off + maps * h * w * bpp].view(maps, h, w, bpp))
        off += maps * h * w * bpp + 4096
    return out


@pytest.mark.parametrize("s,ub,h,w,wins", _PLANE_CASES)
def test_local_mi_operand_planes_forward_by_product_and_backward(s, ub, h, w, wins):
    """(1) miseg_iic_local_joint_fwd_heads_planes returns the joint of miseg_iic_local_joint_fwd_heads bit for bit and leaves, inside the
    windows, exactly the planes miseg_iic_local_make_planes writes (f16 hi, e4m3 of the scaled residual and of the scaled value,
    classes 20..23 zero) -- checked against a torch restatement of the split as well.  (2) miseg_iic_local_bwd_heads_planes on those
    planes equals miseg_iic_local_bwd_heads at precision f16f8 bit for bit (same operand bytes, same MFMA order), store and
    accumulate forms, ragged strips, windows inside the image (zero padding at WINDOW edges), rows and columns outside them untouched."""
    from miseg_amd import _cabi
    k, pad, t, P = 20, 3, 7, len(wins)
    torch.manual_seed(7)
    probs = torch.randn(s, 2 * ub, k, h, w, device=DEV).mul_(2.0).softmax(2).contiguous()
    win = torch.tensor(wins, dtype=torch.int32, device=DEV).view(P, 4)
    st = torch.cuda.current_stream().cuda_stream
    jws = torch.empty(_cabi.query("miseg_iic_local_joint_ws_bytes", ub, k, h, w, pad, P * s), dtype=torch.uint8, device=DEV)
    pb = _cabi.query("miseg_iic_local_planes_bytes", s, ub, k, h, w, pad)
    assert pb == s * 2 * ub * h * w * 88 + 4 * 4096
    assert _cabi.query("miseg_iic_local_planes_bytes", s, ub, k, h, w, 1) == 0 and _cabi.query("miseg_iic_local_planes_bytes", s, ub, 10, h, w, pad) == 0
    ref = torch.full((pb,), 0x7F, dtype=torch.uint8, device=DEV)         # 0x7F = NaN in e4m3 / f16 halves: gaps show
    _cabi.call("miseg_iic_local_make_planes", st, probs.data_ptr(), s, ub, k, h, w, pad, ref.data_ptr(), pb)
    planes = torch.full((pb,), 0x7F, dtype=torch.uint8, device=DEV)
    raw0 = torch.full((s, P, t, t, k, k), float("nan"), device=DEV)
    raw1 = torch.full_like(raw0, float("nan"))
    _cabi.call("miseg_iic_local_joint_fwd_heads", st, probs.data_ptr(), s, ub, k, h, w, pad, win.data_ptr(), P, raw0.data_ptr(), jws.data_ptr(), jws.numel(), 3)
    _cabi.call("miseg_iic_local_joint_fwd_heads_planes", st, probs.data_ptr(), s, ub, k, h, w, pad, win.data_ptr(), P, raw1.data_ptr(), jws.data_ptr(),
               jws.numel(), planes.data_ptr(), pb)
    assert torch.equal(raw0, raw1)
    inside = torch.zeros(h, w, dtype=torch.bool, device=DEV)
    for h0, h1, w0, w1 in wins:
        inside[h0:h1, w0:w1] = True
    maps = s * 2 * ub
    for got, want in zip(_plane_views(planes, maps, h, w), _plane_views(ref, maps, h, w)):
        assert torch.equal(got[:, inside], want[:, inside])
    # the split itself, restated: hi = f16(v); the 8-bit planes hold e4m3(2^20 (v - hi)) and e4m3(2^8 v)
    p16, p8l, p8h = _plane_views(ref, maps, h, w)
    v = probs.view(maps, k, h, w).permute(0, 2, 3, 1).contiguous()
    hi = v.to(torch.float16)
    assert torch.equal(p16.view(torch.float16), hi)
    assert torch.equal(p8l[..., :k].view(torch.float8_e4m3fn).float(), ((v - hi.float()) * 2.0 ** 20).to(torch.float8_e4m3fn).float())
    assert torch.equal(p8h[..., :k].view(torch.float8_e4m3fn).float(), (v * 256.0).to(torch.float8_e4m3fn).float())
    assert int(p8l[..., k:].max()) == 0 and int(p8h[..., k:].max()) == 0
    # backward
    graw = torch.randn(s, P, t, t, k, k, device=DEV)
    scale = torch.rand(s, P, device=DEV) + 0.5
    bws = torch.empty(_cabi.query("miseg_iic_local_bwd_ws_bytes", k, pad, P * s), dtype=torch.uint8, device=DEV)
    for accumulate in (0, 1):
        seed = torch.randn_like(probs)
        outs = []
        for use_planes in (False, True):
            gprob = seed.clone()
            if use_planes:
                _cabi.call("miseg_iic_local_bwd_heads_planes", st, planes.data_ptr(), pb, s, ub, k, h, w, pad, win.data_ptr(), P, graw.data_ptr(),
                           scale.data_ptr(), gprob.data_ptr(), accumulate, bws.data_ptr(), bws.numel())
            else:
                _cabi.call("miseg_iic_local_bwd_heads", st, probs.data_ptr(), s, ub, k, h, w, pad, win.data_ptr(), P, graw.data_ptr(), scale.data_ptr(),
                           gprob.data_ptr(), accumulate, 3, bws.data_ptr(), bws.numel())
            outs.append(gprob)
        assert torch.equal(outs[0], outs[1])
        assert torch.equal(outs[1][..., ~inside], seed[..., ~inside])          # nothing outside the windows is written
        assert not torch.equal(outs[1][..., inside], seed[..., inside])
    with pytest.raises(_cabi.MisegError):
        _cabi.call("miseg_iic_local_bwd_heads_planes", st, planes.data_ptr(), pb - 1, s, ub, k, h, w, pad, win.data_ptr(), P, graw.data_ptr(),
                   scale.data_ptr(), gprob.data_ptr(), 0, bws.data_ptr(), bws.numel())


def test_local_mi_heads_node_keeps_planes_and_matches_the_probability_path():
    """ops.local_mi_heads at f16f8: with the operand planes (default) the node saves the plane buffer instead of probs and its loss and
    input gradient are those of the path without planes, bit for bit."""
    o = ops()
    prev, prev_planes = o.mi_precision_name(), o._MI_PLANES
    o.set_mi_precision("f16f8")
    try:
        torch.manual_seed(3)
        s, ub, k, h, w, pad = 2, 2, 20, 40, 72, 3
        base = torch.randn(s, 2 * ub, k, h, w, device=DEV).softmax(2)
        res = []
        for planes in (False, True):
            o._MI_PLANES = planes
            probs = base.clone().requires_grad_(True)
            loss = o.local_mi_heads(probs, ub, pad, [(0, h, 0, w)])
            saved = loss.grad_fn.saved_tensors[0]
            assert (saved.dtype == torch.uint8) == planes
            g, = torch.autograd.grad(loss.sum(), probs)
            res.append((loss.detach().clone(), g))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    finally:
        o._MI_PLANES = prev_planes
        o.set_mi_precision(prev)


# ------------------------------------------------------------------------------------------ head variants (mlp / normalize)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("head_type,normalize", [("mlp", False), ("mlp", True), ("linear", True)])
def test_local_head_variants_vs_golden_and_oracle(golden, head_type, normalize, dtype):
    """LocalClusterHead with head_type='mlp' and/or normalize=True (ref contrastyou/trainer/_utils.py:137-168) through the module
    (state_dict keys of the reference), forward against the reference's own outputs (heads.npz dec_*), forward + backward incl.
    the fused gather / flip replay against oracle autograd on the dtype-rounded feature."""
    from contrastyou.trainer._utils import LocalClusterHead
    from oracle import losses as OL
    g = golden("heads")
    tag = f"{head_type}_norm{int(normalize)}"
    sd = OH.init_local_cluster_head(8, 6, 3, head_type, seed=6)
    head = LocalClusterHead(input_dim=8, head_type=head_type, num_clusters=6, num_subheads=3, T=1, normalize=normalize)
    head.load_state_dict(sd)
    head = head.to(DEV)
    feat = T(synth.normal(f"dec_{tag}/feat", (3, 8, 10, 12)))
    fd = feat.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-3)
    outs = head(fd)
    assert len(outs) == 3 and outs[0].shape == (3, 6, 10, 12)
    if dtype == torch.float32:
        for s in range(3):
            np.testing.assert_allclose(outs[s].detach().cpu().numpy(), g[f"dec_{tag}/out{s}"], **tol)
    order, dec = [2, 0, 1], [[True, False], [False, True], [True, True]]
    f_ref = fd.detach().float().cpu().contiguous().requires_grad_(True)
    sd_ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_probs = OH.local_cluster_head(sd_ref, OL.apply_flips(torch.stack([f_ref[i] for i in order]), dec), normalize=normalize)
    cot = T(synth.normal("headvar/cot", (3, 3, 6, 10, 12)))
    sum((p * c).sum() for p, c in zip(ref_probs, cot)).backward()
    prob = head.forward_gathered(fd, torch.tensor(order, dtype=torch.int32, device=DEV), ops().flips_to_tensor(dec, DEV))
    for s in range(3):
        np.testing.assert_allclose(prob[s].detach().cpu().numpy(), ref_probs[s].detach().numpy(), **tol)
    (prob * cot.to(DEV)).sum().backward()
    gt = dict(rtol=1e-4, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    for k, p in head.state_dict(keep_vars=True).items():
        ref = sd_ref[k].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=gt["rtol"], atol=gt["atol"] * max(1.0, float(np.abs(ref).max())), err_msg=k)
    np.testing.assert_allclose(fd.grad.float().cpu().numpy(), f_ref.grad.numpy(), rtol=gt["rtol"], atol=gt["atol"] * float(f_ref.grad.abs().max()))


@pytest.mark.parametrize("head_type,normalize", [("mlp", False), ("mlp", True), ("linear", True)])
def test_global_head_variants_vs_golden_and_oracle(golden, head_type, normalize):
    """ClusterHead variants (ref _utils.py:96-134): avg-pool -> Linear(C,128) -> LeakyReLU -> Linear(128,K) [-> L2 normalise] -> softmax."""
    from contrastyou.trainer._utils import ClusterHead
    g = golden("heads")
    tag = f"{head_type}_norm{int(normalize)}"
    sd = OH.init_cluster_head(32, 6, 3, head_type, seed=5)
    head = ClusterHead(input_dim=32, num_clusters=6, num_subheads=3, head_type=head_type, T=1, normalize=normalize)
    head.load_state_dict(sd)
    head = head.to(DEV)
    feat = T(synth.normal(f"enc_{tag}/feat", (5, 32, 6, 6)))
    fd = feat.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    outs = head(fd)
    for s in range(3):
        np.testing.assert_allclose(outs[s].detach().cpu().numpy(), g[f"enc_{tag}/out{s}"], rtol=1e-5, atol=1e-7)
    f_ref = feat.clone().requires_grad_(True)
    sd_ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = OH.cluster_head(sd_ref, f_ref, normalize=normalize)
    cot = T(synth.normal("gheadvar/cot", (3, 5, 6)))
    sum((p * c).sum() for p, c in zip(ref, cot)).backward()
    (torch.stack(outs) * cot.to(DEV)).sum().backward()
    for k, p in head.state_dict(keep_vars=True).items():
        r = sd_ref[k].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), r, rtol=1e-4, atol=2e-6 * max(1.0, float(np.abs(r).max())), err_msg=k)
    np.testing.assert_allclose(fd.grad.cpu().numpy(), f_ref.grad.numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("symmetric", [True, False])
def test_compute_joint_symmetric_switch(symmetric):
    """compute_joint(symmetric=...) (ref iic_loss.py:74-94), value and gradient against the oracle's einsum form in fp64."""
    from contrastyou.losses.iic_loss import compute_joint
    xs, ys = synth.peaked_pair("cj", (24, 7))
    x, y = T(xs).to(DEV).requires_grad_(True), T(ys).to(DEV).requires_grad_(True)
    p = compute_joint(x, y, symmetric=symmetric)
    x64, y64 = T(xs).double().requires_grad_(True), T(ys).double().requires_grad_(True)
    ref = OI.global_joint(x64, y64, symmetric=symmetric)
    np.testing.assert_allclose(p.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-8)
    assert (abs(float((p - p.t()).abs().max())) < 1e-8) == symmetric
    cot = T(synth.normal("cj/cot", (7, 7)))
    (p * cot.to(DEV)).sum().backward()
    (ref * cot.double()).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), x64.grad.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(y.grad.cpu().numpy(), y64.grad.numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("s,ub,h,w,pad,patch", [(5, 16, 256, 256, 3, 1024), (5, 16, 256, 256, 1, 1024), (2, 1, 512, 512, 3, 128)])
def test_joint_checksum_at_full_size(s, ub, h, w, pad, patch, precision):
    """BASELINE full sizes (cfg2: 5 sub-heads x 16 pairs of 20 x 256 x 256 maps, whole-map window, pad 3 and pad 1; cfg4: 512 x 512,
    49 windows of 128 x 128) through the batched launch the bench uses, checked by a size-independent property: the K probabilities
    of a pixel sum to one, so the K x K entries of displacement (a, b) must sum to the NUMBER OF PIXEL PAIRS that displacement has
    inside the window, N (Hw - |a - pad|) (Ww - |b - pad|) -- for every sub-head, window and displacement; and swapping the two
    views transposes the classes and mirrors the displacement (ref iic_loss.py:120-123: conv2d of x with y as the kernel)."""
    from miseg_amd import _cabi
    from contrastyou.losses.iic_loss import _windows
    k, t = 20, 2 * pad + 1
    wins = _windows(h, w, (patch, patch), (patch // 2, patch // 2)) if patch < h else [(0, h, 0, w)]
    P = len(wins)
    gen = torch.Generator(device="cpu").manual_seed(s * 100 + h + pad)
    probs = torch.randn(s, 2 * ub, k, h, w, generator=gen).softmax(2).to(DEV)
    win = torch.tensor(wins, dtype=torch.int32, device=DEV).view(P, 4)
    prec = {"fp32": 0, "bf16x3": 1}[precision]

    def joint(pr):
        raw = torch.empty(s, P, t, t, k, k, device=DEV)
        nb = max(_cabi.query("miseg_iic_local_joint_ws_bytes", ub, k, h, w, pad, P * s), _cabi.query("miseg_iic_local_joint_ws_bytes", ub, k, h, w, pad, P))
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        _cabi.call("miseg_iic_local_joint_fwd_heads", torch.cuda.current_stream().cuda_stream, pr.data_ptr(), s, ub, k, h, w, pad, win.data_ptr(), P,
                   raw.data_ptr(), ws.data_ptr(), ws.numel(), prec)
        return raw
    raw = joint(probs)
    got = raw.double().sum((-1, -2)).cpu()                                   # [S, P, T, T]
    d = (torch.arange(t) - pad).abs().double()
    for p_, (h0, h1, w0, w1) in enumerate(wins):
        want = ub * torch.outer((h1 - h0) - d, (w1 - w0) - d)
        err = float(((got[:, p_] - want).abs() / want).max())
        assert err <= 2e-5, (p_, err)
    # the two views swapped: raw'[a][b][i][j] = raw[T-1-a][T-1-b][j][i]
    swapped = joint(torch.cat([probs[:, ub:], probs[:, :ub]], 1).contiguous())
    mirror = raw.flip(2, 3).transpose(-1, -2)
    scale = float(raw.abs().max())
    assert float((swapped - mirror).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16f8"])
@pytest.mark.parametrize("s,ub,h,w,pad,quad", [(5, 16, 256, 256, 3, False), (5, 16, 256, 256, 1, False), (3, 4, 256, 256, 3, True)])
def test_backward_neighbour_count_at_full_size(s, ub, h, w, pad, quad, precision):
    """The local-MI backward at BASELINE cfg2's full size through the batched launch, checked by a closed form: with dLoss/draw = 1
    everywhere the gradient of a probability is the sum over the displacement window of the OTHER view's class sum = the number of
    in-window neighbours, cnt(row) x cnt(col), identical for every class, sample and sub-head and for both views -- which walks every
    boundary of the kernel's work split (strip edges, the run boundaries between waves and blocks, window edges).  `quad`: four
    non-overlapping 128 x 128 windows in one launch."""
    from miseg_amd import _cabi
    k, t = 20, 2 * pad + 1
    wins = [(0, 128, 0, 128), (0, 128, 128, 256), (128, 256, 0, 128), (128, 256, 128, 256)] if quad else [(0, h, 0, w)]
    P = len(wins)
    gen = torch.Generator(device="cpu").manual_seed(s * 10 + pad)
    probs = torch.randn(s, 2 * ub, k, h, w, generator=gen).softmax(2).to(DEV)
    win = torch.tensor(wins, dtype=torch.int32, device=DEV).view(P, 4)
    graw = torch.ones(s, P, t, t, k, k, device=DEV)
    scale = torch.full((s, P), 0.5, device=DEV)
    gprob = torch.full_like(probs, float("nan"))
    nb = _cabi.query("miseg_iic_local_bwd_ws_bytes", k, pad, P * s)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    _cabi.call("miseg_iic_local_bwd_heads", torch.cuda.current_stream().cuda_stream, probs.data_ptr(), s, ub, k, h, w, pad, win.data_ptr(), P,
               graw.data_ptr(), scale.data_ptr(), gprob.data_ptr(), 0, {"fp32": 0, "bf16x3": 1, "f16f8": 3}[precision], ws.data_ptr(), ws.numel())
    want = torch.zeros(h, w, dtype=torch.float64)
    for h0, h1, w0, w1 in wins:
        r, c = torch.arange(h0, h1), torch.arange(w0, w1)
        cr = (torch.minimum(r + pad, torch.tensor(h1 - 1)) - torch.maximum(r - pad, torch.tensor(h0)) + 1).double()
        cc = (torch.minimum(c + pad, torch.tensor(w1 - 1)) - torch.maximum(c - pad, torch.tensor(w0)) + 1).double()
        want[h0:h1, w0:w1] = 0.5 * torch.outer(cr, cc)
    got = gprob.double().cpu()
    assert bool(torch.isfinite(got).all())
    err = ((got - want) .abs() / want).amax()
    assert float(err) <= 2e-5, float(err)
