"""CPU: pin the oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Tolerances are stated per test; fp64 cases are tight."""
import numpy as np
import pytest
import torch

import synth
from oracle import heads as OH
from oracle import iic as OI
from oracle import losses as OL
from oracle import unet as OU

T = torch.from_numpy

GLOBAL_CASES = [(7, 5), (16, 20)]
LOCAL_CASES = [(3, 5, 12, 10, 2), (4, 20, 32, 32, 1), (4, 20, 32, 32, 3), (2, 8, 64, 64, 3)]
PATCH_CASES = [(2, 4, 100, 100, 1, 32, False), (2, 4, 100, 100, 1, 32, True), (2, 6, 64, 64, 2, 1024, False),
               (1, 3, 512, 512, 3, 128, False), (2, 5, 48, 40, 1, 16, True)]


def tol(tag):
    # fp32: 1e-5 relative to the summand scale (losses are O(1e-3) differences of O(1) entropies,
    # so they get an absolute floor of 1e-6); fp64: 1e-11.
    return (dict(rtol=1e-5, atol=2e-7), np.float32) if tag == "f32" else (dict(rtol=1e-10, atol=1e-13), np.float64)


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("n,k", GLOBAL_CASES)
def test_global_mi(golden, tag, n, k):
    g = golden("iic")
    t, dt = tol(tag)
    key = f"global_{tag}_n{n}_k{k}"
    x = T(synth.probs(key + "/x", (n, k), dt)).requires_grad_(True)
    y = T(synth.probs(key + "/y", (n, k), dt)).requires_grad_(True)
    loss, loss_nl, p = OI.iid_loss(x, y)
    gx, gy = torch.autograd.grad(loss, [x, y])
    np.testing.assert_allclose(p.detach().numpy(), g[f"{key}/joint"], **t)
    np.testing.assert_allclose(float(loss), float(g[f"{key}/loss"]), rtol=t["rtol"], atol=t["atol"] * 10)
    np.testing.assert_allclose(float(loss_nl), float(g[f"{key}/loss_no_lamb"]), rtol=t["rtol"], atol=t["atol"] * 10)
    np.testing.assert_allclose(gx.numpy(), g[f"{key}/gx"], rtol=t["rtol"] * 10, atol=t["atol"] * 10)
    np.testing.assert_allclose(gy.numpy(), g[f"{key}/gy"], rtol=t["rtol"] * 10, atol=t["atol"] * 10)


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("n,k,h,w,p", LOCAL_CASES)
def test_local_mi(golden, tag, n, k, h, w, p):
    g = golden("iic")
    t, dt = tol(tag)
    key = f"local_{tag}_n{n}_k{k}_h{h}_w{w}_p{p}"
    x = T(synth.probs(key + "/x", (n, k, h, w), dt)).requires_grad_(True)
    y = T(synth.probs(key + "/y", (n, k, h, w), dt)).requires_grad_(True)
    raw = OI.local_joint_raw(x, y, p)                       # [T,T,K,K]
    ref_raw = np.transpose(g[f"{key}/raw_kktt"], (2, 3, 0, 1))  # reference conv output is [K,K,T,T]
    np.testing.assert_allclose(raw.detach().numpy(), ref_raw, rtol=t["rtol"], atol=t["atol"] * raw.detach().abs().max().item())
    loss = OI.local_mi_from_raw(raw)
    np.testing.assert_allclose(float(loss), float(g[f"{key}/loss"]), rtol=t["rtol"], atol=t["atol"] * 10)
    gx, gy = torch.autograd.grad(loss, [x, y])
    gscale = float(np.abs(synth.fp_unpack(g, f"{key}/gx")["sample"]).max())
    synth.check_fingerprint(gx.numpy(), synth.fp_unpack(g, f"{key}/gx"), f"{key}/gx", rtol=t["rtol"] * 20, atol=gscale * t["rtol"] * 20)
    synth.check_fingerprint(gy.numpy(), synth.fp_unpack(g, f"{key}/gy"), f"{key}/gy", rtol=t["rtol"] * 20, atol=gscale * t["rtol"] * 20)


@pytest.mark.parametrize("n,k,h,w,p", LOCAL_CASES[:3])
def test_local_mi_closed_form_gradient(n, k, h, w, p):
    """dLoss/dR emitted by the HIP epilogue == autograd through the epilogue (fp64)."""
    key = f"local_f64_n{n}_k{k}_h{h}_w{w}_p{p}"
    x = T(synth.probs(key + "/x", (n, k, h, w), np.float64))
    y = T(synth.probs(key + "/y", (n, k, h, w), np.float64))
    raw = OI.local_joint_raw(x, y, p).requires_grad_(True)
    (auto,) = torch.autograd.grad(OI.local_mi_from_raw(raw), [raw])
    closed = OI.local_mi_grad_wrt_raw(raw.detach())
    np.testing.assert_allclose(closed.numpy(), auto.numpy(), rtol=1e-9, atol=1e-12 * float(auto.abs().max()) + 1e-18)


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("n,k,h,w,p,patch,use_mask", PATCH_CASES)
def test_patch_local_mi(golden, tag, n, k, h, w, p, patch, use_mask):
    """fp64 pins the algorithm (tight).  In fp32 the MI gradient of near-independent inputs is a
    difference of O(1) logs that nearly cancels, so the reference's OWN fp32 gradients carry
    rounding noise of a few % of their scale; fp32 is therefore compared at 5 % of the gradient scale."""
    g = golden("iic")
    dt = np.float32 if tag == "f32" else np.float64
    key = f"patch_{tag}_n{n}_k{k}_h{h}_w{w}_p{p}_ps{patch}_m{int(use_mask)}"
    x = T(synth.probs(key + "/x", (n, k, h, w), dt)).requires_grad_(True)
    y = T(synth.probs(key + "/y", (n, k, h, w), dt)).requires_grad_(True)
    m = T(synth.mask(key + "/mask", (n, 1, h, w)).astype(dt)) if use_mask else None
    loss = OI.iid_seg_small_patch_loss(x, y, p, patch, mask=m)
    if tag == "f64":
        np.testing.assert_allclose(float(loss), float(g[f"{key}/loss"]), rtol=1e-9, atol=1e-13)
    else:
        np.testing.assert_allclose(float(loss), float(g[f"{key}/loss"]), rtol=1e-5, atol=2e-6)
    gx, gy = torch.autograd.grad(loss, [x, y])
    gscale = float(np.abs(synth.fp_unpack(g, f"{key}/gx")["sample"]).max())
    rt = 1e-7 if tag == "f64" else 5e-2
    for nm, val in (("gx", gx), ("gy", gy)):
        synth.check_fingerprint(val.numpy(), synth.fp_unpack(g, f"{key}/{nm}"), f"{key}/{nm}", rtol=rt, atol=gscale * rt)


PEAK_GLOBAL = [(16, 20), (64, 5)]
PEAK_LOCAL = [(3, 5, 12, 10, 2), (4, 20, 32, 32, 1), (4, 20, 32, 32, 3), (2, 8, 64, 64, 3)]
PEAK_PATCH = [(2, 4, 100, 100, 1, 32), (1, 20, 96, 96, 3, 32)]


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_peaked_inputs_literal_relative_tolerance(golden, tag):
    """Correlated, peaked inputs (synth.peaked_pair): the losses are O(0.1 .. 1), so north_star's '1e-5 relative' is held literally
    (fp64: 1e-10) for all three losses; gradients at 1e-4 of their scale (fp64: 1e-8)."""
    g = golden("iic")
    dt = np.float32 if tag == "f32" else np.float64
    lrt, grt = (1e-5, 1e-4) if tag == "f32" else (1e-10, 1e-8)
    for n, k in PEAK_GLOBAL:
        key = f"gpeak_{tag}_n{n}_k{k}"
        xs, ys = synth.peaked_pair(key, (n, k), dt)
        x, y = T(xs).requires_grad_(True), T(ys).requires_grad_(True)
        loss, loss_nl, p = OI.iid_loss(x, y)
        assert abs(float(g[f"{key}/loss"])) > 0.1
        np.testing.assert_allclose(float(loss), float(g[f"{key}/loss"]), rtol=lrt)
        np.testing.assert_allclose(float(loss_nl), float(g[f"{key}/loss_no_lamb"]), rtol=lrt)
        np.testing.assert_allclose(p.detach().numpy(), g[f"{key}/joint"], rtol=lrt, atol=lrt * 1e-3)
        gx, gy = torch.autograd.grad(loss, [x, y])
        sc = float(np.abs(g[f"{key}/gx"]).max())
        np.testing.assert_allclose(gx.numpy(), g[f"{key}/gx"], rtol=grt, atol=grt * sc)
        np.testing.assert_allclose(gy.numpy(), g[f"{key}/gy"], rtol=grt, atol=grt * sc)
    for n, k, h, w, p_ in PEAK_LOCAL:
        key = f"lpeak_{tag}_n{n}_k{k}_h{h}_w{w}_p{p_}"
        xs, ys = synth.peaked_pair(key, (n, k, h, w), dt)
        x, y = T(xs).requires_grad_(True), T(ys).requires_grad_(True)
        loss = OI.iid_seg_loss(x, y, p_)
        assert abs(float(g[f"{key}/loss"])) > 0.1
        np.testing.assert_allclose(float(loss), float(g[f"{key}/loss"]), rtol=lrt)
        gx, gy = torch.autograd.grad(loss, [x, y])
        sc = float(np.abs(synth.fp_unpack(g, f"{key}/gx")["sample"]).max())
        for nm, val in (("gx", gx), ("gy", gy)):
            synth.check_fingerprint(val.numpy(), synth.fp_unpack(g, f"{key}/{nm}"), f"{key}/{nm}", rtol=grt, atol=sc * grt)
    for n, k, h, w, p_, patch in PEAK_PATCH:
        key = f"ppeak_{tag}_n{n}_k{k}_h{h}_w{w}_p{p_}_ps{patch}"
        xs, ys = synth.peaked_pair(key, (n, k, h, w), dt)
        x, y = T(xs).requires_grad_(True), T(ys).requires_grad_(True)
        loss = OI.iid_seg_small_patch_loss(x, y, p_, patch)
        np.testing.assert_allclose(float(loss), float(g[f"{key}/loss"]), rtol=lrt)
        gx, gy = torch.autograd.grad(loss, [x, y])
        sc = float(np.abs(synth.fp_unpack(g, f"{key}/gx")["sample"]).max())
        for nm, val in (("gx", gx), ("gy", gy)):
            synth.check_fingerprint(val.numpy(), synth.fp_unpack(g, f"{key}/{nm}"), f"{key}/{nm}", rtol=grt, atol=sc * grt)


@pytest.mark.parametrize("h,patch", [(100, 32), (64, 1024), (512, 128), (224, 1024), (48, 16), (33, 16)])
def test_patch_geometry(golden, h, patch):
    wins = OI.patch_windows(h, h, (patch, patch), (patch // 2, patch // 2))
    np.testing.assert_array_equal(np.asarray(wins, dtype=np.int64), golden("iic")[f"patchgeom_h{h}_ps{patch}/windows"])


@pytest.mark.parametrize("head_type", ["linear", "mlp"])
@pytest.mark.parametrize("normalize", [False, True])
def test_heads(golden, head_type, normalize):
    g = golden("heads")
    tag = f"{head_type}_norm{int(normalize)}"
    outs = OH.cluster_head(OH.init_cluster_head(32, 6, 3, head_type, seed=5), T(synth.normal(f"enc_{tag}/feat", (5, 32, 6, 6))),
                           normalize=normalize)
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.numpy(), g[f"enc_{tag}/out{i}"], rtol=1e-5, atol=1e-7)
    outs = OH.local_cluster_head(OH.init_local_cluster_head(8, 6, 3, head_type, seed=6),
                                 T(synth.normal(f"dec_{tag}/feat", (3, 8, 10, 12))), normalize=normalize)
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.numpy(), g[f"dec_{tag}/out{i}"], rtol=1e-5, atol=1e-7)


def test_unet_train_eval(golden):
    g = golden("unet")
    sd = OU.init_state(1, 4, seed=3)
    x = T(synth.uniform("unet/x64", (3, 1, 64, 64)))
    wgt = T(synth.normal("unet/w64", (3, 4, 64, 64)))
    names = OU.trainable_keys(sd)
    for k in names:
        sd[k].requires_grad_(True)
    logits, feats = OU.unet_forward(sd, x, training=True)
    np.testing.assert_allclose(logits.detach().numpy(), g["train64/logits"], rtol=1e-4, atol=1e-5)
    for name in ("Conv1", "Conv2", "Conv3", "Conv4", "Conv5", "Up_conv5", "Up_conv4", "Up_conv3", "Up_conv2"):
        synth.check_fingerprint(feats[name].detach().numpy(), synth.fp_unpack(g, f"train64/feat/{name}"),
                                f"train64/feat/{name}", rtol=1e-4, atol=1e-5)
    (logits * wgt).sum().backward()
    for k in names:
        fp = synth.fp_unpack(g, f"train64/grad/{k}")
        scale = float(np.abs(fp["sample"]).max()) + 1e-12
        synth.check_fingerprint(sd[k].grad.numpy(), fp, f"train64/grad/{k}", rtol=2e-3, atol=2e-3 * scale)
    for k in sd:
        if "running" in k or "num_batches" in k:
            np.testing.assert_allclose(sd[k].detach().numpy(), g[f"train64/after/{k}"], rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        ev, _ = OU.unet_forward(sd, x, training=False)
    np.testing.assert_allclose(ev.numpy(), g["eval64/logits"], rtol=1e-4, atol=1e-5)


def test_unet_256(golden):
    sd = OU.init_state(1, 4, seed=3)
    with torch.no_grad():
        logits, _ = OU.unet_forward(sd, T(synth.uniform("unet/x256", (2, 1, 256, 256))), training=True, update_stats=False)
    synth.check_fingerprint(logits.numpy(), synth.fp_unpack(golden("unet"), "train256/logits"), "train256/logits",
                            rtol=1e-4, atol=1e-5)


def test_kl_mse_onehot_simplex(golden):
    g = golden("losses")
    logits = T(synth.normal("kl/logits", (3, 4, 16, 16))).requires_grad_(True)
    target = T(synth.integers("kl/target", (3, 16, 16), 4))
    onehot = OL.class2one_hot(target, 4)
    assert onehot.dtype == torch.int64
    np.testing.assert_array_equal(onehot.numpy(), g["kl/onehot"])          # integer: bit-exact
    kl = OL.kl_div(logits.softmax(1), onehot)
    (gl,) = torch.autograd.grad(kl, [logits])
    np.testing.assert_allclose(float(kl), float(g["kl/loss"]), rtol=1e-6)
    np.testing.assert_allclose(gl.numpy(), g["kl/glogits"], rtol=1e-5, atol=1e-9)
    a = T(synth.normal("mse/a", (3, 4, 16, 16))).requires_grad_(True)
    b = T(synth.normal("mse/b", (3, 4, 16, 16)))
    mse = OL.softmax_mse(a, b)
    (ga,) = torch.autograd.grad(mse, [a])
    np.testing.assert_allclose(float(mse), float(g["mse/loss"]), rtol=1e-6)
    np.testing.assert_allclose(ga.numpy(), g["mse/ga"], rtol=1e-5, atol=1e-10)
    cases = T(g["simplex/cases"])
    assert [OI.simplex(c) for c in cases] == [bool(v) for v in g["simplex/is_simplex"]]
    assert [OL.is_one_hot(c) for c in cases] == [bool(v) for v in g["simplex/is_one_hot"]]


@pytest.mark.parametrize("seed", [0, 123, 9999999, 4242])
def test_flip_replay(golden, seed):
    g = golden("losses")
    x = torch.arange(4 * 2 * 3 * 5, dtype=torch.float32).view(4, 2, 3, 5)
    dec = OL.flip_decisions(seed, 4)
    np.testing.assert_array_equal(np.asarray(dec), g[f"flip/seed{seed}/decisions"])
    np.testing.assert_array_equal(OL.apply_flips(x, dec).numpy(), g[f"flip/seed{seed}/out"])  # bit-exact data movement


def test_dice_avg_sched(golden):
    g = golden("meters_sched")
    meter = OL.DiceMeter(4, report_axis=[1, 2, 3])
    for it in range(3):
        meter.add(T(synth.integers(f"dice/pred{it}", (4, 12, 12), 4)), T(synth.integers(f"dice/target{it}", (4, 12, 12), 4)),
                  group_name=[str(s) for s in g["dice/groups"][it]])
    summ = meter.summary()
    assert list(summ.keys()) == [str(k) for k in g["dice/keys"]]
    np.testing.assert_allclose(list(summ.values()), g["dice/values"], rtol=1e-6)
    for max_epoch, warm, mult in ((100, 10, 400), (30, 5, 300)):
        np.testing.assert_allclose(OL.warmup_cosine_lrs(1e-7, mult, warm, max_epoch),
                                   g[f"sched_e{max_epoch}_w{warm}_m{mult}/lrs"], rtol=1e-9)


# ----------------------------------------------------------------------------- full step
from oracle import step as OS  # noqa: E402

STEP = dict(H=64, LB=2, UB=3, NB=2, lr=1e-3, wd=1e-5, cons_weight=5.0, iic_weight=0.1)


def step_inputs(mode):
    """Same synthetic state/batches make_golden.step_inputs feeds the reference."""
    H, LB, UB, NB = (STEP[k] for k in ("H", "LB", "UB", "NB"))
    model_sd = OU.init_state(1, 4, seed=9)
    heads = {"Conv5": OH.init_cluster_head(256, 20, 5, "linear", seed=10),
             "Up_conv3": OH.init_local_cluster_head(32, 20, 5, "linear", seed=11),
             "Up_conv2": OH.init_local_cluster_head(16, 20, 5, "linear", seed=12)}
    lab = [(T(synth.uniform(f"step/{mode}/lab{i}", (LB, 1, H, H))), T(synth.integers(f"step/{mode}/tgt{i}", (LB, 1, H, H), 4)))
           for i in range(NB)]
    unl = [T(synth.uniform(f"step/{mode}/unl{i}", (UB, 1, H, H))) for i in range(NB)]
    return model_sd, heads, lab, unl


REF_HEAD_PREFIX = {"Conv5": "_encoder_projectors._clusters.Conv5.", "Up_conv3": "_decoder_projectors._clusters.Up_conv3.",
                   "Up_conv2": "_decoder_projectors._clusters.Up_conv2."}


def ref_param_name(oracle_name):
    """oracle 'Up_conv3/_headers.0.0.weight' -> reference optimizer name 'proj/_decoder_projectors...'."""
    if "/" not in oracle_name:
        return oracle_name
    f, k = oracle_name.split("/", 1)
    return "proj/" + REF_HEAD_PREFIX[f] + k


@pytest.mark.parametrize("mode", ["udaiic", "partial", "uda", "iic"])
def test_full_step(golden, mode):
    """Two optimiser steps of the reference's UDAIICEpocher / TrainEpocher / UDATrainEpocher / IICTrainEpocher
    (semi_seg/epocher.py:110-323) vs the oracle restatement: every meter, every parameter gradient of step 1, every parameter
    after step 2."""
    g = golden("step")
    model_sd, heads, lab, unl = step_inputs(mode)
    has_heads = mode in ("udaiic", "iic")
    state = OS.StepState(model_sd, heads if has_heads else {}, lr=STEP["lr"], weight_decay=STEP["wd"])
    if not has_heads:
        state.heads = {}
    seeds = [int(s) for s in g[f"{mode}/seeds"]]
    import random
    random.seed(1234)
    assert [random.randint(0, int(1e7)) for _ in seeds] == seeds      # epocher.py:144 draw sequence
    logs, grads1 = [], None
    dice = OL.DiceMeter(4, report_axis=[1, 2, 3])
    for i in range(STEP["NB"]):
        sc, gr = OS.train_step(state, lab[i][0], lab[i][1], unl[i], seeds[i], mode=mode,
                               feature_importance=[float(v) for v in g[f"{mode}/feature_importance"]],
                               cons_weight=STEP["cons_weight"], iic_weight=STEP["iic_weight"])
        dice.add(sc.pop("pred"), lab[i][1].squeeze(1), group_name=[f"patient{j:03d}_00" for j in range(STEP["LB"])])
        logs.append(sc)
        grads1 = grads1 or gr
    ref = dict(zip([str(k) for k in g[f"{mode}/meter_keys"]], g[f"{mode}/meter_values"]))
    mean = lambda k: float(np.mean([l[k] for l in logs]))  # noqa: E731
    np.testing.assert_allclose(mean("sup_loss"), ref["sup_loss/mean"], rtol=1e-5)
    np.testing.assert_allclose(mean("reg_loss"), ref["reg_loss/mean"], rtol=1e-4, atol=1e-7)
    for k, v in dice.summary().items():
        np.testing.assert_allclose(v, ref[f"sup_dice/{k}"], rtol=1e-6)
    if mode in ("udaiic", "uda"):
        np.testing.assert_allclose(mean("uda"), ref["uda/mean"], rtol=1e-4)
    if mode in ("udaiic", "iic"):
        np.testing.assert_allclose(mean("mi"), ref["mi/mean"], rtol=1e-3, atol=2e-6)
        for f in ("Conv5", "Up_conv3", "Up_conv2"):
            np.testing.assert_allclose(mean(f"mi/{f}"), ref[f"individual_mis/{f}"], rtol=1e-3, atol=2e-6)
    for name, gr in grads1.items():
        fp = synth.fp_unpack(g, f"{mode}/grad_step1/{ref_param_name(name)}")
        scale = float(np.abs(fp["sample"]).max()) + 1e-12
        # global-MI head gradients are O(1e-8) here (MI of the untrained Conv5 head ~ 2e-6): pure fp32
        # rounding noise in the reference itself, hence the 1e-9 absolute floor.
        synth.check_fingerprint(gr.numpy(), fp, f"{mode}/grad_step1/{ref_param_name(name)}", rtol=5e-3,
                                atol=max(5e-3 * scale, 1e-9))
    for k, v in state.model.items():
        fp = synth.fp_unpack(g, f"{mode}/model_after/{k}")
        # Adam's first steps move every weight by ~lr regardless of gradient size, so a gradient that is
        # rounding noise (sign-unstable) can move a weight by 2*lr; bound the check by that.
        synth.check_fingerprint(v.detach().numpy(), fp, f"{mode}/model_after/{k}", rtol=1e-4, atol=2.5 * STEP["lr"] * STEP["NB"])
