"""GPU parity of the WHOLE train step through the reference-compatible surface: my UDAIICEpocher / TrainEpocher
(HIP kernels, exact-fp32 mode) vs the meters, gradients and updated weights the reference's own epochers produced
on the same synthetic state and batches (tests/golden/step.npz)."""
import os
import random
import sys

import numpy as np
import pytest
import torch

import synth
from oracle import heads as OH
from oracle import unet as OU

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda"
STEP = dict(H=64, LB=2, UB=3, NB=2, lr=1e-3, wd=1e-5, cons_weight=5.0, iic_weight=0.1)
FEATURES = ["Conv5", "Up_conv3", "Up_conv2"]


def build(mode, dtype="float32"):
    from contrastyou.arch import UNet
    from deepclustering2.loss import KL_div
    from deepclustering2.optim import Adam
    from semi_seg._utils import IICLossWrapper, ProjectorWrapper
    from itertools import chain
    H, LB, UB, NB = (STEP[k] for k in ("H", "LB", "UB", "NB"))
    model = UNet(1, 4, compute_dtype=dtype)
    model.load_state_dict(OU.init_state(1, 4, seed=9))
    pw = ProjectorWrapper()
    pw.init_encoder(feature_names=FEATURES, num_clusters=20, num_subheads=5, head_types="linear", normalize=False)
    pw.init_decoder(feature_names=FEATURES, num_clusters=20, num_subheads=5, head_types="linear", normalize=False)
    pw._encoder_projectors["Conv5"].load_state_dict(OH.init_cluster_head(256, 20, 5, "linear", seed=10))
    pw._decoder_projectors["Up_conv3"].load_state_dict(OH.init_local_cluster_head(32, 20, 5, "linear", seed=11))
    pw._decoder_projectors["Up_conv2"].load_state_dict(OH.init_local_cluster_head(16, 20, 5, "linear", seed=12))
    lw = IICLossWrapper(feature_names=FEATURES, paddings=[1, 3], patch_sizes=1024)
    model, pw = model.to(DEV), pw.to(DEV)
    params = chain(model.parameters(), pw.parameters()) if mode in ("udaiic", "iic") else model.parameters()     # semi_seg/trainer.py:150-184
    opt = Adam(params, lr=STEP["lr"], weight_decay=STEP["wd"])

    def loader(tag, B, with_tgt):
        for i in range(NB):
            img = T(synth.uniform(f"step/{mode}/{tag}{i}", (B, 1, H, H)))
            tgt = T(synth.integers(f"step/{mode}/tgt{i}", (B, 1, H, H), 4)) if with_tgt else torch.zeros(B, 1, H, H, dtype=torch.long)
            yield [[[img, tgt], [img.clone(), tgt.clone()]], [f"patient{j:03d}_00_{j}" for j in range(B)], ["0"] * B,
                   [f"patient{j:03d}_00" for j in range(B)]]

    return model, pw, lw, opt, loader("lab", LB, True), loader("unl", UB, False), KL_div(verbose=False)


def make_epocher(mode, model, pw, lw, opt, lab, unl, kl, fi, num_batches):
    """The epocher of `Trainer.name = mode` (semi_seg/trainer.py:132-195), constructed as the trainers do."""
    from semi_seg.epocher import IICTrainEpocher, TrainEpocher, UDAIICEpocher, UDATrainEpocher
    if mode == "udaiic":
        return UDAIICEpocher(model, pw, opt, lab, unl, kl, torch.nn.MSELoss(), lw, num_batches=num_batches, cur_epoch=0, device=DEV,
                             feature_position=FEATURES, feature_importance=fi, cons_weight=STEP["cons_weight"], iic_weight=STEP["iic_weight"])
    if mode == "uda":
        return UDATrainEpocher(model, opt, lab, unl, kl, torch.nn.MSELoss(), STEP["cons_weight"], num_batches, 0, DEV,
                               feature_position=FEATURES, feature_importance=fi)
    if mode == "iic":
        return IICTrainEpocher(model, pw, opt, lab, unl, kl, lw, STEP["iic_weight"], num_batches, 0, DEV, feature_position=FEATURES,
                               feature_importance=fi)
    return TrainEpocher(model, opt, lab, unl, kl, 0, num_batches, 0, DEV, feature_position=FEATURES, feature_importance=fi)


MODES = ["udaiic", "partial", "uda", "iic"]      # BASELINE configs[1] / configs[0]; SURVEY 8(f-4): semi_seg/epocher.py:200-284


@pytest.mark.parametrize("mode", MODES)
def test_epocher_matches_reference_run(golden, mode):
    g = golden("step")
    model, pw, lw, opt, lab, unl, kl = build(mode)
    fi = [float(v) for v in g[f"{mode}/feature_importance"]]
    random.seed(1234)
    res = make_epocher(mode, model, pw, lw, opt, lab, unl, kl, fi, STEP["NB"]).run()
    got = {f"{k}/{kk}": float(vv) for k, v in res.items() for kk, vv in dict(v).items()}
    ref = dict(zip([str(k) for k in g[f"{mode}/meter_keys"]], g[f"{mode}/meter_values"]))
    assert set(got) == set(ref), set(got) ^ set(ref)                      # identical meter names
    np.testing.assert_allclose(got["lr/mean"], ref["lr/mean"], rtol=1e-12)
    # The golden meters are means over 2 iterations.  Iteration 2 runs on weights after one Adam step, and Adam moves
    # every weight by ~lr whatever its gradient size: parameters whose gradient is rounding noise step +-lr by the
    # sign of that noise, so iteration-2 losses of two correct implementations differ at the 1e-3..1e-1 level
    # (UDA most).  Hence: loose bounds on the 2-iteration means here, tight bounds on iteration 1 below.
    np.testing.assert_allclose(got["sup_loss/mean"], ref["sup_loss/mean"], rtol=3e-3)
    for k in ("sup_dice/DSC1", "sup_dice/DSC2", "sup_dice/DSC3", "sup_dice/DSC_mean"):
        np.testing.assert_allclose(got[k], ref[k], rtol=2e-2, atol=5e-3)   # rare classes: a few pixels move a 1e-2 Dice
    np.testing.assert_allclose(got["reg_loss/mean"], ref["reg_loss/mean"], rtol=0.2, atol=1e-7)
    if mode in ("udaiic", "uda"):
        np.testing.assert_allclose(got["uda/mean"], ref["uda/mean"], rtol=0.2)
    if mode in ("udaiic", "iic"):
        for k in ("mi/mean", "individual_mis/Conv5", "individual_mis/Up_conv3", "individual_mis/Up_conv2"):
            np.testing.assert_allclose(got[k], ref[k], rtol=0.2, atol=5e-6)
    if mode == "udaiic":
        assert got["iic_weight/mean"] == ref["iic_weight/mean"] and got["uda_weight/mean"] == ref["uda_weight/mean"]
    # Weights after two Adam steps.  Adam moves every weight by ~lr per step whatever the gradient size, so comparing the weights
    # themselves says nothing about the gradients (a wrong-sign gradient would still land within 2*lr): the gradients are compared
    # directly in test_step_gradients_match_reference.  What IS checked here is the update DIRECTION: wherever the reference moved a
    # weight by more than 1.2*lr over the two steps (both steps agreed in sign), this run must have moved it the same way.
    init = OU.init_state(1, 4, seed=9)
    agree, total = 0, 0
    for k, v in model.state_dict().items():
        if not k.endswith(("weight", "bias")):
            continue
        fp = synth.fp_unpack(g, f"{mode}/model_after/{k}")
        idx = synth.sample_index(v.numel(), f"{mode}/model_after/{k}")
        mine = v.detach().float().cpu().numpy().reshape(-1).astype(np.float64)[idx] - init[k].numpy().reshape(-1).astype(np.float64)[idx]
        ref_move = fp["sample"] - init[k].numpy().reshape(-1).astype(np.float64)[idx]
        sel = np.abs(ref_move) > 1.2 * STEP["lr"]
        agree += int((np.sign(mine[sel]) == np.sign(ref_move[sel])).sum())
        total += int(sel.sum())
        assert np.abs(mine).max() <= 2.5 * STEP["lr"] * STEP["NB"], k
    assert total > 1000 and agree >= 0.97 * total, (agree, total)


def test_udaiic_step_bf16_runs_and_tracks_fp32(golden):
    """bf16 compute mode (BASELINE cfg2 dtype): same step, losses within bf16 tolerance of the fp32 golden."""
    from semi_seg.epocher import UDAIICEpocher
    g = golden("step")
    model, pw, lw, opt, lab, unl, kl = build("udaiic", "bfloat16")
    fi = [float(v) for v in g["udaiic/feature_importance"]]
    random.seed(1234)
    res = UDAIICEpocher(model, pw, opt, lab, unl, kl, torch.nn.MSELoss(), lw, num_batches=STEP["NB"], cur_epoch=0, device=DEV,
                        feature_position=FEATURES, feature_importance=fi, cons_weight=5.0, iic_weight=0.1).run()
    ref = dict(zip([str(k) for k in g["udaiic/meter_keys"]], g["udaiic/meter_values"]))
    np.testing.assert_allclose(res["sup_loss"]["mean"], ref["sup_loss/mean"], rtol=2e-2)
    np.testing.assert_allclose(res["uda"]["mean"], ref["uda/mean"], rtol=0.25)
    assert abs(res["mi"]["mean"] - ref["mi/mean"]) < 0.5 * abs(ref["mi/mean"]) + 1e-4


@pytest.mark.parametrize("mode", MODES)
def test_first_iteration_is_tight(golden, mode):
    """Iteration 1 (identical weights on both sides): every meter of the HIP epocher vs the oracle step, which is
    itself pinned to the reference run by tests/test_oracle_golden.py::test_full_step."""
    from oracle import step as OS
    from oracle import losses as OL
    g = golden("step")
    model, pw, lw, opt, lab, unl, kl = build(mode)
    fi = [float(v) for v in g[f"{mode}/feature_importance"]]
    random.seed(1234)
    res = make_epocher(mode, model, pw, lw, opt, lab, unl, kl, fi, 1).run()
    H, LB, UB = STEP["H"], STEP["LB"], STEP["UB"]
    heads = {"Conv5": OH.init_cluster_head(256, 20, 5, "linear", seed=10), "Up_conv3": OH.init_local_cluster_head(32, 20, 5, "linear", seed=11),
             "Up_conv2": OH.init_local_cluster_head(16, 20, 5, "linear", seed=12)}
    state = OS.StepState(OU.init_state(1, 4, seed=9), heads if mode in ("udaiic", "iic") else {}, lr=STEP["lr"], weight_decay=STEP["wd"])
    limg = T(synth.uniform(f"step/{mode}/lab0", (LB, 1, H, H)))
    ltgt = T(synth.integers(f"step/{mode}/tgt0", (LB, 1, H, H), 4))
    uimg = T(synth.uniform(f"step/{mode}/unl0", (UB, 1, H, H)))
    sc, _ = OS.train_step(state, limg, ltgt, uimg, int(g[f"{mode}/seeds"][0]), mode=mode, feature_importance=fi,
                          cons_weight=STEP["cons_weight"], iic_weight=STEP["iic_weight"], do_update=False)
    np.testing.assert_allclose(res["sup_loss"]["mean"], sc["sup_loss"], rtol=2e-5)            # north-star 1e-5 class
    np.testing.assert_allclose(res["reg_loss"]["mean"], sc["reg_loss"], rtol=2e-4, atol=1e-7)
    dice = OL.DiceMeter(4, report_axis=[1, 2, 3])
    dice.add(sc["pred"], ltgt.squeeze(1), group_name=[f"patient{j:03d}_00" for j in range(LB)])
    for k, v in dice.summary().items():
        np.testing.assert_allclose(res["sup_dice"][k], v, rtol=2e-3)   # integer counts; one argmax tie may move a pixel
    if mode in ("udaiic", "uda"):
        np.testing.assert_allclose(res["uda"]["mean"], sc["uda"], rtol=2e-4)
    if mode in ("udaiic", "iic"):
        np.testing.assert_allclose(res["mi"]["mean"], sc["mi"], rtol=2e-3, atol=2e-6)
        for f in FEATURES:
            np.testing.assert_allclose(res["individual_mis"][f], sc[f"mi/{f}"], rtol=2e-3, atol=2e-6)


def _first_iteration_gradients(mode, dtype, monkeypatch, wgrad_side=True, iic_side=True):
    """Run iteration 1 of the epocher and return {reference parameter name: gradient} as the optimiser kernel sees them: the flat
    gradient buffer is cloned between FlatBuffers.collect() and miseg_adam_step (the only place all slots are final)."""
    from miseg_amd import unet_ops
    monkeypatch.setattr(unet_ops, "_WGRAD_SIDE", wgrad_side)
    monkeypatch.setenv("MISEG_IIC_STREAM", "1" if iic_side else "0")
    model, pw, lw, opt, lab, unl, kl = build(mode, dtype)
    grabbed = []
    real = unet_ops.adam_step

    def spy(param, grad, *a, **k):
        grabbed.append(grad.detach().clone())
        return real(param, grad, *a, **k)

    monkeypatch.setattr(unet_ops, "adam_step", spy)
    random.seed(1234)
    make_epocher(mode, model, pw, lw, opt, lab, unl, kl, [0.5, 0.25, 0.25], 1).run()
    assert len(grabbed) == 1
    flat, fb = grabbed[0].cpu() / float(opt.grad_scale), opt.flat      # fp16 mode: the kernel divides the static loss scale out
    named = list(model.named_parameters()) + ([("proj/" + n, p) for n, p in pw.named_parameters()] if mode in ("udaiic", "iic") else [])
    out = {}
    for name, p in named:
        o = fb.offset_of(p)
        out[name] = flat[o:o + p.numel()].view(p.shape).numpy()
    return out


def _gradient_errors(g, mode, grads):
    """Per parameter: relative L2 error over the fingerprint sample, and max |error| relative to the tensor's largest entry."""
    names = [str(n) for n in g[f"{mode}/param_names"]]
    assert sorted(names) == sorted(grads), set(names) ^ set(grads)
    rows = {}
    for n in names:
        fp = synth.fp_unpack(g, f"{mode}/grad_step1/{n}")
        got = grads[n].reshape(-1).astype(np.float64)[synth.sample_index(grads[n].size, f"{mode}/grad_step1/{n}")]
        ref = fp["sample"]
        rows[n] = (float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30)), float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)),
                   float(np.abs(ref).max()))
    return rows


def _dump(tag, rows):
    """Achieved errors go to gpurun_out/ (merged back from the GPU box) so DESIGN.md can quote them."""
    import json
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f"step_grad_errors_{tag}.json"), "w") as f:
        json.dump({k: {"rel_l2": v[0], "max_rel_to_scale": v[1], "scale": v[2]} for k, v in rows.items()}, f, indent=1)


@pytest.mark.parametrize("wgrad_side,iic_side", [(True, True), (False, False)])
@pytest.mark.parametrize("mode", MODES)
def test_step_gradients_match_reference(golden, monkeypatch, mode, wgrad_side, iic_side):
    """Every parameter gradient of iteration 1 -- through the flat-gradient slots, the wgrad side stream, the IIC side stream, the
    symbolic loss seeding and the partial zero-fill of the tap gradient -- against the gradients the REFERENCE's own epocher handed
    to its optimiser (step.npz `{mode}/grad_step1/*`, recorded by tests/golden/make_golden.py::RecordingAdam), with the side
    streams on (shipped) and off.  fp32 mode.  Bounds: the last decoder block, the logits layer and the decoder-tap heads see at
    most two ReLUs / no BatchNorm statistics downstream and must be tight; deeper layers inherit ReLU-mask flips (see
    test_unet_fp32_vs_golden) and are bounded in relative L2."""
    g = golden("step")
    grads = _first_iteration_gradients(mode, "float32", monkeypatch, wgrad_side, iic_side)
    rows = _gradient_errors(g, mode, grads)
    _dump(f"{mode}_fp32_w{int(wgrad_side)}i{int(iic_side)}", rows)
    worst = {k: v[0] for k, v in rows.items()}
    # measured (round 2, gpurun_out/step_grad_errors_*): last block / logits layer 2e-7 .. 3e-4 (median 7e-6), decoder-tap heads
    # <= 3e-5, everything behind more ReLUs 1e-3 .. 1.2e-2; the bounds below leave ~3x of that
    bad = {k: v for k, v in worst.items() if v > 3e-2 and rows[k][2] > 1e-7}     # <= 1e-7: the global-MI head's gradients are fp32 noise
    assert not bad, bad
    tail = sorted(v for k, v in worst.items() if k.startswith(("Up_conv2", "DeConv")))
    # The last block's errors are decided by WHICH activations of `Up_conv2.conv` sit within rounding distance of zero in a given batch
    # (a flipped ReLU mask moves that layer's BatchNorm bias gradient by ~1e-3 relative on this fixture's 8 K pixels), i.e. by the last
    # bits of the batch statistics.  Measured on the four fixture batches (gpurun_out/step_grad_errors_*), median / worst of the block:
    #   statistics as a float tree + finalize launch (round 3, MISEG_BN_ACC=0): udaiic 7e-6 / 3e-4, partial 8e-6 / 4e-4, uda 1.4e-4 / 4.9e-3, iic 2.8e-3 / 4.6e-3
    #   statistics as fixed-point accumulators (round 4, shipped; exacter): udaiic 1.6e-3 / 3.2e-3, partial 4e-6 / 3.8e-4, uda 3e-5 / 5.7e-3, iic 4e-6 / 2.9e-4
    # -- the same envelope, dealt differently to the batches (the CPU oracle shows it against the reference too:
    # tests/test_oracle_golden.py::test_full_step bounds it at 5e-3 of each tensor's scale).  One bound for all four modes; the logits
    # layer and the block's tightest tensor are flip-free in every mode and must be tight everywhere.
    med, top = 5e-3, 1.5e-2
    assert tail[len(tail) // 2] < med and tail[0] < 1e-5 and tail[-1] < top, tail
    assert max(v for k, v in worst.items() if k.startswith("DeConv")) < 2e-5, worst
    if mode in ("udaiic", "iic"):
        dec_heads = sorted(v for k, v in worst.items() if "_decoder_projectors" in k)
        assert dec_heads[-1] < 2e-4, dec_heads


def _oracle_names(g, mode):
    """oracle parameter name ('<tap>/_headers...') -> reference optimiser name ('proj/_decoder_projectors._clusters.<tap>...')."""
    pref = {"Conv5": "proj/_encoder_projectors._clusters.Conv5.", "Up_conv3": "proj/_decoder_projectors._clusters.Up_conv3.",
            "Up_conv2": "proj/_decoder_projectors._clusters.Up_conv2."}

    def to_ref(n):
        if "/" not in n:
            return n
        f, k = n.split("/", 1)
        return pref[f] + k
    return to_ref


@pytest.mark.parametrize("mi_precision", ["f16f8", "bf16x3"])
def test_step_gradients_bf16_match_bf16_emulation_and_track_reference(golden, monkeypatch, mi_precision):
    """The bench's arithmetic (bf16 activations and activation gradients, bf16x3 local MI), iteration-1 gradients of every
    parameter against
      (a) the CPU oracle step with the SAME rounding points (oracle.unet.unet_forward_bf16_autograd: straight-through bf16
          roundings of every stored tensor): only accumulation order and ReLU-mask / pool-tie flips differ -> must be close;
      (b) the reference's fp32 gradients (step.npz): on this random-init net, forward bf16 rounding alone moves the gradient of
          Up_conv2.conv.3 by 18 % and of Conv1 by 80 % (the CPU emulation shows the same profile: scratch analysis recorded in
          DESIGN.md section 2) -- so (b) only guards against gross errors and records the achieved figures."""
    from miseg_amd import ops as _ops
    from oracle import step as OS
    g = golden("step")
    _ops.set_mi_precision(mi_precision)       # f16f8 = what `Arch.compute_dtype=bfloat16` and bench.py run by default
    try:
        grads = _first_iteration_gradients("udaiic", "bfloat16", monkeypatch)
    finally:
        _ops.set_mi_precision("fp32")
    rows = _gradient_errors(g, "udaiic", grads)
    _dump(f"udaiic_bf16_{mi_precision}_vs_fp32_reference", rows)
    sig = {k: v[0] for k, v in rows.items() if v[2] > 1e-7}
    assert max(sig.values()) < 1.3, {k: v for k, v in sig.items() if v >= 1.3}
    assert sig["DeConv_1x1.weight"] < 3e-2 and sig["Up_conv2.conv.3.weight"] < 0.3
    # (a) same rounding points on the CPU
    H, LB, UB = STEP["H"], STEP["LB"], STEP["UB"]
    heads = {"Conv5": OH.init_cluster_head(256, 20, 5, "linear", seed=10), "Up_conv3": OH.init_local_cluster_head(32, 20, 5, "linear", seed=11),
             "Up_conv2": OH.init_local_cluster_head(16, 20, 5, "linear", seed=12)}
    state = OS.StepState(OU.init_state(1, 4, seed=9), heads, lr=STEP["lr"], weight_decay=STEP["wd"])
    limg = T(synth.uniform("step/udaiic/lab0", (LB, 1, H, H)))
    ltgt = T(synth.integers("step/udaiic/tgt0", (LB, 1, H, H), 4))
    uimg = T(synth.uniform("step/udaiic/unl0", (UB, 1, H, H)))
    _, emu = OS.train_step(state, limg, ltgt, uimg, int(g["udaiic/seeds"][0]), mode="udaiic", cons_weight=STEP["cons_weight"],
                           iic_weight=STEP["iic_weight"], do_update=False, unet_fn=OU.unet_forward_bf16_autograd)
    to_ref = _oracle_names(g, "udaiic")
    emu_rows = {}
    for n, ge in emu.items():
        mine, ref = grads[to_ref(n)].astype(np.float64), ge.numpy().astype(np.float64)
        emu_rows[to_ref(n)] = (float(np.linalg.norm(mine - ref) / (np.linalg.norm(ref) + 1e-30)),
                               float(np.abs(mine - ref).max() / (np.abs(ref).max() + 1e-30)), float(np.abs(ref).max()))
    _dump(f"udaiic_bf16_{mi_precision}_vs_bf16_emulation", emu_rows)
    sig = {k: v[0] for k, v in emu_rows.items() if v[2] > 1e-7}
    # Measured (round 2): logits layer 2.6e-3, decoder-tap heads 3.5e-3 / 1.9e-2, Up_conv2.conv.3 7e-2, then growing steadily to
    # 0.48 at Conv1 -- two bf16 evaluations with different accumulation order decorrelate with depth on a RANDOM-INIT ReLU+BatchNorm
    # stack (perturbations grow ~1.25x per layer there: the fp32 logits of this net already move by 17 % RMS under bf16 rounding),
    # so only the layers next to the losses can be held tightly.
    assert max(sig.values()) < 0.8, {k: v for k, v in sig.items() if v >= 0.8}
    assert sig["DeConv_1x1.weight"] < 1e-2 and sig["DeConv_1x1.bias"] < 1e-2, (sig["DeConv_1x1.weight"], sig["DeConv_1x1.bias"])
    assert sig["Up_conv2.conv.3.weight"] < 0.15, sig["Up_conv2.conv.3.weight"]
    dec = sorted(v for k, v in sig.items() if "_decoder_projectors" in k)
    assert dec[-1] < 6e-2 and dec[len(dec) // 2] < 2e-2, dec


def test_deferred_readback_records_every_iteration_like_the_synchronous_one():
    """The loop reads iteration i's scalars and Dice counts after iteration i+1 is enqueued (pinned ring, semi_seg/epocher.py
    `_after_step`).  An epoch must give exactly the meters of the per-iteration read-back (MISEG_DEFER_FETCH=0 behaviour), and
    a NaN loss must still end the epoch with the reference's RuntimeError (ref semi_seg/epocher.py:129-130)."""
    from semi_seg.epocher import UDAIICEpocher
    results = []
    for defer in (True, False):
        model, pw, lw, opt, lab, unl, kl = build("udaiic")
        random.seed(99)
        ep = UDAIICEpocher(model, pw, opt, lab, unl, kl, torch.nn.MSELoss(), lw, num_batches=STEP["NB"], cur_epoch=0, device=DEV,
                           feature_position=FEATURES, feature_importance=[0.5, 0.25, 0.25], cons_weight=5.0, iic_weight=0.1)
        ep._DEFER_FETCH = defer
        results.append((repr(dict(ep.run())), opt.flat.flat_param.detach().clone()))
        assert ep._inflight is None
    assert results[0][0] == results[1][0]
    assert torch.equal(results[0][1], results[1][1])
    model, pw, lw, opt, lab, unl, kl = build("udaiic")
    with torch.no_grad():
        list(pw.parameters())[-1].fill_(float("nan"))   # a head bias: every probability of that tap is NaN -> not a simplex
    ep = UDAIICEpocher(model, pw, opt, lab, unl, kl, torch.nn.MSELoss(), lw, num_batches=STEP["NB"], cur_epoch=0, device=DEV,
                       feature_position=FEATURES, feature_importance=[0.5, 0.25, 0.25], cons_weight=5.0, iic_weight=0.1)
    opt.flat.ensure()      # the flat buffers are built lazily: force them, then snapshot the parameters
    snap = opt.flat.flat_param.detach().clone()
    with pytest.raises((RuntimeError, AssertionError)):
        ep.run()
    # the failed check is raised one iteration late, but it guarded the Adam launches on the device: no weight has moved -- the state
    # the reference, which raises before backward (iic_loss.py:147-148), leaves behind.  (NaN-free part of the buffer: the poisoned bias.)
    if os.environ.get("MISEG_GUARD_STEP", "1") != "0":
        now = opt.flat.flat_param.detach()
        keep = ~torch.isnan(snap)
        assert torch.equal(now[keep], snap[keep]) and bool(torch.isnan(now[~keep]).all())


def _tape_run(mode_dtype, mi_precision, tape: bool, steps: int = 7, trainer: str = "udaiic"):
    """`steps` iterations of the bench's step at a small shape; with the launch tape: 2 eager, 1 recorded, the rest replayed."""
    import random
    import bench
    from miseg_amd import ops
    ops.set_mi_precision(mi_precision)
    torch.manual_seed(0)
    random.seed(7)
    ep, opt = bench.build_step(torch.device("cuda"), 2, 2, 64, mode_dtype, 0)
    ep._TAPE_DEFAULT = False
    drv = bench.StepDriver(ep)
    if tape:
        ep.enable_step_tape(warmup=2)
    random.seed(11)
    for _ in range(steps):
        drv.step()
    drv.close()
    tp = ep._step_tape
    info = None if tp is None else (tp.replays, tp.disabled, tp.n_ops, bool(tp.handle))
    out = opt.flat.flat_param.detach().clone(), opt._m[0].detach().clone(), dict(ep.meters.tracking_status()), opt._steps[0], info
    ep.disable_step_tape()
    ops.set_mi_precision("fp32")
    return out


@pytest.mark.parametrize("dtype,mi", [("float32", "fp32"), ("bfloat16", "f16f8"), ("float16", "f16f8")])
def test_step_tape_replay_equals_eager_steps(dtype, mi):
    """The launch tape (miseg_amd.tape.StepTape: the library records its own entry-point calls during one eager iteration and issues
    them again from one C call) must reproduce eager iterations bit for bit: same parameters, same Adam moments, same meters after a
    mix of eager warm-up, the recorded iteration and replays with fresh batches and fresh flip decisions -- and the shipped trainer
    must record clean (no ATen launch outside the library), i.e. the tape must really have been replayed."""
    p_t, m_t, met_t, n_t, info = _tape_run(dtype, mi, True)
    p_e, m_e, met_e, n_e, _ = _tape_run(dtype, mi, False)
    assert info is not None and info[1] is None, f"the tape was refused: {info}"
    assert info[3] and info[0] == 4, f"expected 2 eager + 1 recorded + 4 replayed iterations: {info}"
    assert n_t == n_e == 7
    assert torch.equal(p_e, p_t), float((p_e - p_t).abs().max())
    assert torch.equal(m_e, m_t), float((m_e - m_t).abs().max())
    assert repr(met_e) == repr(met_t), (met_e, met_t)     # repr: the unused lr meter is nan in both


def test_tape_guard_sees_the_backward_thread():
    """The recording's safety net: the ForeignOps dispatch mode must see ATen operators issued from autograd's worker thread (where
    the backward halves of the step run), or a launch outside the library would be silently missing from every replay."""
    from miseg_amd.tape import ForeignOps

    class Leaky(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.view_as(x)

        @staticmethod
        def backward(ctx, g):
            return g + 1.0            # an ATen launch on the engine's thread

    x = torch.ones(8, device="cuda", requires_grad=True)
    guard = ForeignOps()
    with guard:
        y = Leaky.apply(x)
        y.backward(torch.ones(8, device="cuda").expand(8))
    assert any(n.startswith("aten::add") for n in guard.seen), guard.seen
    guard = ForeignOps()
    with guard:
        v = torch.empty(4, 4, device="cuda").view(16)[2:6].detach()
        torch.zeros(3)                # host tensors do not count
    assert guard.seen == [], guard.seen


def test_fp16_dynamic_loss_scale_under_the_tape():
    """fp16 storage mode under the launch tape: the loss scale lives in the step block (seeds = coefficient x scale, hyper[4] = 1 /
    scale), so an overflow during REPLAYED iterations still skips the update on the device and halves the scale on the host."""
    import random
    import bench
    from miseg_amd import ops
    from miseg_amd.flat import LossScaler
    ops.set_mi_precision("f16f8")
    torch.manual_seed(0)
    random.seed(3)
    ep, opt = bench.build_step(torch.device("cuda"), 2, 2, 64, "float16", 0)
    ep._TAPE_DEFAULT = False
    drv = bench.StepDriver(ep)
    ep.enable_step_tape(warmup=2)
    try:
        for _ in range(4):               # 2 eager, 1 recorded, 1 replayed at the default scale
            drv.step()
        ep._flush_records()
        tp = ep._step_tape
        assert tp.disabled is None and tp.replays == 1, (tp.disabled, tp.replays)
        scaler = opt.loss_scaler
        assert isinstance(scaler, LossScaler) and scaler.overflows == 0
        before = opt.flat.flat_param.detach().clone()
        scaler.scale = 2.0 ** 40         # every half-precision activation gradient overflows
        with pytest.warns(UserWarning, match="overflow"):
            for _ in range(3):
                drv.step()
            ep._flush_records()
        assert tp.replays == 4
        assert torch.equal(opt.flat.flat_param.detach(), before)        # three skipped updates
        # three overflowed (skipped) iterations; the scale halves once per overflow AT THE CURRENT SCALE: the iteration staged before the
        # first count arrived overflowed at the old scale and does not halve again (flat.LossScaler.update)
        assert scaler.scale == 2.0 ** 38 and scaler.overflows == 3
        assert opt._steps[0] == 4                                       # skipped updates do not count towards the bias corrections
        scaler.scale = 16384.0
        for _ in range(2):
            drv.step()
        ep._flush_records()
        assert not torch.equal(opt.flat.flat_param.detach(), before)    # updates resume at a sane scale
    finally:
        drv.close()
        ep.disable_step_tape()
        ops.set_mi_precision("fp32")


def test_bench_line_contract():
    """`python bench.py` prints ONE JSON line with the driver's fields, the roofline object of the dominant kernel and the CPU
    baseline object (small shape here; the default invocation is the BASELINE configuration)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "2", "--lb", "2", "--ub", "2", "--size", "64"],
                         cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 4 * 1000.0 / d["ms_per_step"]) < 0.02 * d["value"]
    rl = d["roofline"]
    assert rl["bound"] in ("hbm", "mfma") and rl["achieved"] > 0 and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "workload" in d["config"]


def test_step_gradients_fp16_track_reference_and_loss_scale_is_neutral(golden, monkeypatch):
    """IEEE-half storage mode (BASELINE cfg5: fp16 with all three MI taps): iteration-1 gradients of every parameter, after the fused
    Adam's un-scaling, against the reference's fp32 gradients.  Half carries 3 more mantissa bits than bf16, so the layers next to
    the losses must sit ~8x closer to fp32 than the bf16 run does (test above: logits layer 3e-2, Up_conv2.conv.3 0.3); deeper
    layers decorrelate on this random-init net as they do in bf16.  Run at two loss scales: the un-scaled gradients must agree
    (scaling by a power of two is exact in half unless a value leaves its range -- so this is the no-overflow / no-underflow
    check), and the meters must not depend on the scale at all."""
    from miseg_amd import ops as _ops
    g = golden("step")
    _ops.set_mi_precision("bf16x3")
    runs = {}
    try:
        for scale in ("16384", "1024"):
            monkeypatch.setenv("MISEG_LOSS_SCALE", scale)
            runs[scale] = _first_iteration_gradients("udaiic", "float16", monkeypatch)
    finally:
        _ops.set_mi_precision("fp32")
    rows = _gradient_errors(g, "udaiic", runs["16384"])
    _dump("udaiic_fp16_vs_fp32_reference", rows)
    sig = {k: v[0] for k, v in rows.items() if v[2] > 1e-7}
    assert all(np.isfinite(v) for v in sig.values())
    assert max(sig.values()) < 1.0, {k: v for k, v in sig.items() if v >= 1.0}
    assert sig["DeConv_1x1.weight"] < 5e-3 and sig["Up_conv2.conv.3.weight"] < 6e-2, (sig["DeConv_1x1.weight"], sig["Up_conv2.conv.3.weight"])
    dec = sorted(v for k, v in sig.items() if "_decoder_projectors" in k)
    assert dec[-1] < 2e-2, dec
    # the two loss scales: same gradients next to the losses (deeper layers inherit the usual flip noise of a re-run)
    for k in ("DeConv_1x1.weight", "DeConv_1x1.bias", "Up_conv2.conv.3.weight"):
        a, b = runs["16384"][k].astype(np.float64), runs["1024"][k].astype(np.float64)
        assert np.linalg.norm(a - b) <= 2e-2 * np.linalg.norm(a), (k, np.linalg.norm(a - b) / np.linalg.norm(a))


def test_adam_scaled_equals_adam_on_unscaled_gradients():
    """miseg_adam_step_scaled(grad * s, s) == miseg_adam_step(grad) for a power-of-two s (exact), and rejects s <= 0."""
    from miseg_amd import _cabi, unet_ops
    n = 4099
    p0 = T(synth.normal("adamsc/p", (n,))).to(DEV)
    gr = T(synth.normal("adamsc/g", (n,), scale=1e-3)).to(DEV)
    hyper = torch.tensor([1e-3 / (1 - 0.9), 1.0 / (1 - 0.999) ** 0.5, 1e-8, 1e-5], device=DEV)
    outs = []
    for s in (1.0, 4096.0):
        p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        unet_ops.adam_step(p, gr * s, m, v, hyper, 0.9, 0.999, s)
        outs.append((p.cpu(), m.cpu(), v.cpu()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    with pytest.raises(_cabi.MisegError):
        unet_ops.adam_step(p0.clone(), gr, torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), hyper, 0.9, 0.999, 0.0)


def test_first_iteration_non_default_configuration_matches_the_oracle(monkeypatch):
    """What config/semi.yaml lets a user change on the regulariser side, all at once, against the CPU oracle step on identical weights
    and inputs: FOUR taps at four scales (Conv5, Up_conv4, Up_conv3, Up_conv2), `mlp` heads with `normalize: true`, 10 clusters x 3
    sub-heads (off the K = 20 bf16 fast path -> the generic fp32-MFMA local-MI kernels and `heads_var.hip`), paddings [2, 1, 3],
    32-pixel overlapping patches, and the `kl` consistency criterion (ref semi_seg/trainer.py:137-160, _utils.py:96-168,
    iic_loss.py:152-189).  fp32 mode: meters at the tolerances of test_first_iteration_is_tight, gradients of every parameter next
    to a loss in relative L2."""
    from oracle import step as OS
    from contrastyou.arch import UNet
    from deepclustering2.loss import KL_div
    from deepclustering2.optim import Adam
    from semi_seg._utils import IICLossWrapper, ProjectorWrapper
    from semi_seg.epocher import UDAIICEpocher
    from itertools import chain
    from miseg_amd import unet_ops
    feats, fi, pads, K, S = ["Conv5", "Up_conv4", "Up_conv3", "Up_conv2"], [1.0, 0.5, 0.5, 0.25], [2, 1, 3], 10, 3
    H, LB, UB = 64, 2, 3
    model = UNet(1, 4, compute_dtype="float32")
    model.load_state_dict(OU.init_state(1, 4, seed=21))
    pw = ProjectorWrapper()
    pw.init_encoder(feature_names=feats, num_clusters=K, num_subheads=S, head_types="mlp", normalize=True)
    pw.init_decoder(feature_names=feats, num_clusters=K, num_subheads=S, head_types="mlp", normalize=True)
    heads = {"Conv5": OH.init_cluster_head(256, K, S, "mlp", seed=30), "Up_conv4": OH.init_local_cluster_head(64, K, S, "mlp", seed=31),
             "Up_conv3": OH.init_local_cluster_head(32, K, S, "mlp", seed=32), "Up_conv2": OH.init_local_cluster_head(16, K, S, "mlp", seed=33)}
    pw._encoder_projectors["Conv5"].load_state_dict(heads["Conv5"])
    for f in feats[1:]:
        pw._decoder_projectors[f].load_state_dict(heads[f])
    lw = IICLossWrapper(feature_names=feats, paddings=pads, patch_sizes=32)
    model, pw = model.to(DEV), pw.to(DEV)
    opt = Adam(chain(model.parameters(), pw.parameters()), lr=1e-3, weight_decay=1e-5)
    limg = T(synth.uniform("stepvar/lab", (LB, 1, H, H)))
    ltgt = T(synth.integers("stepvar/tgt", (LB, 1, H, H), 4))
    uimg = T(synth.uniform("stepvar/unl", (UB, 1, H, H)))

    def loader(img, tgt):
        B = img.shape[0]
        yield [[[img, tgt], [img.clone(), tgt.clone()]], [f"patient{j:03d}_00_{j}" for j in range(B)], ["0"] * B, [f"patient{j:03d}_00" for j in range(B)]]

    grabbed, real = [], unet_ops.adam_step

    def spy(param, grad, *a, **k):
        grabbed.append(grad.detach().clone())
        return real(param, grad, *a, **k)
    monkeypatch.setattr(unet_ops, "adam_step", spy)
    random.seed(4321)
    ep = UDAIICEpocher(model, pw, opt, loader(limg, ltgt), loader(uimg, torch.zeros(UB, 1, H, H, dtype=torch.long)), KL_div(verbose=False),
                       KL_div(verbose=False), lw, num_batches=1, cur_epoch=0, device=DEV, feature_position=feats, feature_importance=fi,
                       cons_weight=5.0, iic_weight=0.1)
    res = ep.run()
    # the flip seed of iteration 1: the epocher draws it with random.randint after random.seed(4321) (ref epocher.py:146)
    random.seed(4321)
    seed = random.randint(0, int(1e7))
    state = OS.StepState(OU.init_state(1, 4, seed=21), heads, lr=1e-3, weight_decay=1e-5)
    sc, grads = OS.train_step(state, limg, ltgt, uimg, seed, mode="udaiic", feature_names=feats, feature_importance=fi, paddings=pads,
                              patch_sizes=[32, 32, 32], cons_weight=5.0, iic_weight=0.1, do_update=False, head_normalize=True,
                              uda_criterion="kl")
    np.testing.assert_allclose(res["sup_loss"]["mean"], sc["sup_loss"], rtol=2e-5)
    np.testing.assert_allclose(res["uda"]["mean"], sc["uda"], rtol=2e-4)
    np.testing.assert_allclose(res["mi"]["mean"], sc["mi"], rtol=2e-3, atol=2e-6)
    for f in feats:
        np.testing.assert_allclose(res["individual_mis"][f], sc[f"mi/{f}"], rtol=2e-3, atol=2e-6)
    np.testing.assert_allclose(res["reg_loss"]["mean"], sc["reg_loss"], rtol=2e-4, atol=1e-7)
    # gradients: the logits layer and every head parameter (the layers next to the losses)
    flat, fb = grabbed[0].cpu(), opt.flat
    pref = {"Conv5": pw._encoder_projectors["Conv5"], "Up_conv4": pw._decoder_projectors["Up_conv4"],
            "Up_conv3": pw._decoder_projectors["Up_conv3"], "Up_conv2": pw._decoder_projectors["Up_conv2"]}
    worst = {}
    for name, g_ref in grads.items():
        if "/" in name:
            f, k = name.split("/", 1)
            p = dict(pref[f].named_parameters())[k]
        elif name.startswith("DeConv_1x1"):
            p = dict(model.named_parameters())[name]
        else:
            continue
        o = fb.offset_of(p)
        mine = flat[o:o + p.numel()].view(p.shape).double()
        ref = g_ref.double()
        if float(ref.abs().max()) > 1e-7:
            worst[name] = float((mine - ref).norm() / ref.norm())
    # measured: logits layer 1e-6, decoder-tap heads <= 2.7e-4, the global head of Conv5 <= 4.5e-3 (its gradient is a difference of
    # nearly equal terms at this near-uniform initialisation: ~1e-6 in size, fp32 cancellation on both sides)
    assert len(worst) >= 4 * S * 2
    loose = {k: v for k, v in worst.items() if k.startswith("Conv5/")}
    tight = {k: v for k, v in worst.items() if not k.startswith("Conv5/")}
    assert max(tight.values()) < 1e-3, {k: v for k, v in tight.items() if v >= 1e-3}
    assert max(loose.values()) < 2e-2, loose


_CFG2_ORACLE = {}


def _cfg2_oracle():
    """ONE oracle step at BASELINE configs[1]'s shape (LB = UB = 16, 256 x 256, default taps / heads / paddings): ~35 s of CPU -- taken
    from tests/golden/bigstep_oracle.npz while the fingerprint of oracle/ + synth.py stored there still matches (tests/golden/bigstep.py),
    recomputed otherwise; shared by the parametrisations below."""
    if not _CFG2_ORACLE:
        import bigstep
        sc, grads = bigstep.result("cfg2", False)
        _CFG2_ORACLE.update(bigstep.cfg2_inputs(), sc=sc, grads=grads)
    return _CFG2_ORACLE


def _cfg2_oracle_bf16():
    """The same step with the U-Net's bf16 rounding points emulated on the CPU (oracle.unet.unet_forward_bf16_autograd: every stored
    activation and activation gradient rounded to bf16, straight through): what the bf16 HIP step is held against -- the fp32 oracle
    differs from any bf16 evaluation of this random-init net by 10-25 % in the consistency term alone."""
    o = _cfg2_oracle()
    if "sc_bf16" not in o:
        import bigstep
        sc, grads = bigstep.result("cfg2", True)
        o.update(sc_bf16=sc, grads_bf16=grads)
    return o


@pytest.mark.parametrize("dtype,mi_precision", [("float32", "fp32"), ("bfloat16", "f16f8"), ("bfloat16", "bf16x3")])
def test_whole_step_at_the_bench_shape_matches_the_oracle(monkeypatch, dtype, mi_precision):
    """BASELINE configs[1] at its OWN size -- LB = UB = 16, 256 x 256, taps Conv5 / Up_conv3 / Up_conv2, 20 clusters x 5 sub-heads,
    paddings [1, 3], whole-map patches -- one `udaiic` iteration (ref semi_seg/epocher.py:137-188, 308-323) of the HIP epocher against
    the CPU oracle on the same weights and batch: sup / uda / mi / per-tap mi / reg_loss and the gradients of the logits layer and of
    every head parameter.  fp32 mode at the tolerances of test_first_iteration_is_tight; bf16 mode (the bench's arithmetic, with
    either matrix-core form of the local-MI contraction) at the tolerances of test_udaiic_step_bf16_runs_and_tracks_fp32, the
    gradients in relative L2."""
    from contrastyou.arch import UNet
    from deepclustering2.loss import KL_div
    from deepclustering2.optim import Adam
    from semi_seg._utils import IICLossWrapper, ProjectorWrapper
    from semi_seg.epocher import UDAIICEpocher
    from itertools import chain
    from miseg_amd import ops as _ops, unet_ops
    o = _cfg2_oracle()
    H, LB, UB = 256, 16, 16
    model = UNet(1, 4, compute_dtype=dtype)
    model.load_state_dict(OU.init_state(1, 4, seed=40))
    pw = ProjectorWrapper()
    pw.init_encoder(feature_names=FEATURES, num_clusters=20, num_subheads=5, head_types="linear", normalize=False)
    pw.init_decoder(feature_names=FEATURES, num_clusters=20, num_subheads=5, head_types="linear", normalize=False)
    pw._encoder_projectors["Conv5"].load_state_dict(o["heads"]["Conv5"])
    pw._decoder_projectors["Up_conv3"].load_state_dict(o["heads"]["Up_conv3"])
    pw._decoder_projectors["Up_conv2"].load_state_dict(o["heads"]["Up_conv2"])
    lw = IICLossWrapper(feature_names=FEATURES, paddings=[1, 3], patch_sizes=1024)
    model, pw = model.to(DEV), pw.to(DEV)
    opt = Adam(chain(model.parameters(), pw.parameters()), lr=1e-3, weight_decay=1e-5)

    def loader(img, tgt):
        B = img.shape[0]
        yield [[[img, tgt], [img.clone(), tgt.clone()]], [f"patient{j:03d}_00_{j}" for j in range(B)], ["0"] * B, [f"patient{j:03d}_00" for j in range(B)]]

    grabbed, real = [], unet_ops.adam_step

    def spy(param, grad, *a, **k):
        grabbed.append(grad.detach().clone())
        return real(param, grad, *a, **k)
    monkeypatch.setattr(unet_ops, "adam_step", spy)
    _ops.set_mi_precision(mi_precision)
    try:
        random.seed(2468)
        res = UDAIICEpocher(model, pw, opt, loader(o["limg"], o["ltgt"]), loader(o["uimg"], torch.zeros(UB, 1, H, H, dtype=torch.long)),
                            KL_div(verbose=False), torch.nn.MSELoss(), lw, num_batches=1, cur_epoch=0, device=DEV, feature_position=FEATURES,
                            feature_importance=[0.5, 0.25, 0.25], cons_weight=5.0, iic_weight=0.1).run()
    finally:
        _ops.set_mi_precision("fp32")
    exact = dtype == "float32"
    sc = o["sc"] if exact else _cfg2_oracle_bf16()["sc_bf16"]       # bf16 storage: against the oracle with the same rounding points
    _dump(f"cfg2shape_scalars_{dtype}_{mi_precision}", {k: (abs(res[k]["mean"] - sc[k]) / (abs(sc[k]) + 1e-30), res[k]["mean"], sc[k])
                                                         for k in ("sup_loss", "uda", "mi", "reg_loss")})
    np.testing.assert_allclose(res["sup_loss"]["mean"], sc["sup_loss"], rtol=2e-5 if exact else 5e-3)
    np.testing.assert_allclose(res["uda"]["mean"], sc["uda"], rtol=2e-4 if exact else 2e-2)
    if exact:
        np.testing.assert_allclose(res["mi"]["mean"], sc["mi"], rtol=2e-3, atol=2e-6)
        for f in FEATURES:
            np.testing.assert_allclose(res["individual_mis"][f], sc[f"mi/{f}"], rtol=2e-3, atol=2e-6)
        np.testing.assert_allclose(res["reg_loss"]["mean"], sc["reg_loss"], rtol=2e-4, atol=1e-7)
    else:
        np.testing.assert_allclose(res["mi"]["mean"], sc["mi"], rtol=5e-2, atol=2e-6)
        for f in FEATURES:
            np.testing.assert_allclose(res["individual_mis"][f], sc[f"mi/{f}"], rtol=5e-2, atol=2e-6)
        np.testing.assert_allclose(res["reg_loss"]["mean"], sc["reg_loss"], rtol=2e-2, atol=1e-6)
    flat, fb = grabbed[0].cpu() / float(opt.grad_scale), opt.flat
    pref = {"Conv5": pw._encoder_projectors["Conv5"], "Up_conv3": pw._decoder_projectors["Up_conv3"], "Up_conv2": pw._decoder_projectors["Up_conv2"]}
    worst = {}
    for name, g_ref in o["grads"].items():
        if "/" in name:
            f, k = name.split("/", 1)
            p = dict(pref[f].named_parameters())[k]
        elif name.startswith("DeConv_1x1"):
            p = dict(model.named_parameters())[name]
        else:
            continue
        off = fb.offset_of(p)
        mine, ref = flat[off:off + p.numel()].view(p.shape).double(), g_ref.double()
        if float(ref.abs().max()) > 1e-7:
            worst[name] = float((mine - ref).norm() / ref.norm())
    _dump(f"cfg2shape_{dtype}_{mi_precision}", {k: (v, v, 0.0) for k, v in worst.items()})
    dec = {k: v for k, v in worst.items() if not k.startswith("Conv5/")}
    assert len(dec) >= 2 + 2 * 2                                   # logits layer (weight, bias) + the two decoder heads' parameters
    if exact:      # measured: logits layer < 1e-7, decoder-tap heads 4e-6 .. 7e-4
        assert max(dec.values()) < 2e-3, {k: v for k, v in dec.items() if v >= 2e-3}
        assert max(v for k, v in dec.items() if k.startswith("DeConv_1x1")) < 1e-5, dec
        assert all(v < 2e-2 for k, v in worst.items() if k.startswith("Conv5/")), worst
    else:       # bf16 features / logits: the layers next to the losses stay within bf16 forward rounding of the fp32 gradients
        # measured (round 3, both matrix-core forms alike): logits layer 1.5e-3 / 9e-4, decoder-tap heads 5e-4 .. 6e-3
        assert max(v for k, v in dec.items() if k.startswith("DeConv_1x1")) < 5e-3, dec
        assert max(dec.values()) < 2e-2, {k: v for k, v in dec.items() if v >= 2e-2}


def test_count_nonfinite_guards_the_fused_adam():
    """`miseg_count_nonfinite` + the guard of `miseg_adam_step_guarded`: a flat gradient with one inf (or NaN) leaves parameters and both
    moments bit-identical; a clean one is counted as 0 and steps."""
    from miseg_amd import unet_ops
    g = torch.Generator().manual_seed(3)
    n = 100_003
    p0 = torch.randn(n, generator=g).to(DEV)
    grad = torch.randn(n, generator=g).to(DEV)
    hyper = torch.tensor([1e-3, 1.0, 1e-8, 1e-5], device=DEV)
    for poison in (None, float("inf"), float("-inf"), float("nan")):
        p, m, v, gr = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), grad.clone()
        if poison is not None:
            gr[n - 7] = poison
            gr[5] = poison
        bad = unet_ops.count_nonfinite(gr)
        unet_ops.adam_step(p, gr, m, v, hyper, 0.9, 0.999, 1.0, bad)
        assert float(bad) == (0.0 if poison is None else 2.0)
        if poison is None:
            assert not torch.equal(p, p0) and float(m.abs().max()) > 0
        else:
            assert torch.equal(p, p0) and float(m.abs().max()) == 0 and float(v.abs().max()) == 0


def test_fp16_overflow_skips_the_step_and_halves_the_loss_scale(monkeypatch):
    """BASELINE configs[4]'s arithmetic (IEEE-half storage) with an absurd initial loss scale (2^40: every activation gradient leaves
    half's range): the overflow count guards the fused Adam on the device -- no weight, no moment moves -- and the scale follows one
    iteration late (flat.LossScaler): three iterations -> three overflows seen, two of them at the then-current scale -> scale 2^38
    (the iteration staged before the first count arrived does not halve it again).  With the default scale the same three
    iterations are clean and move the weights."""
    import warnings
    from semi_seg.epocher import UDAIICEpocher
    for init, expect_overflows in ((2.0 ** 40, 3), (None, 0)):
        if init is not None:
            monkeypatch.setenv("MISEG_LOSS_SCALE", repr(init))
        else:
            monkeypatch.delenv("MISEG_LOSS_SCALE", raising=False)
        STEP3 = dict(STEP)
        model, pw, lw, opt, lab, unl, kl = build("udaiic", "float16")
        before = {k: v.detach().clone() for k, v in model.state_dict().items()}

        def cyc(gen):
            items = list(gen)
            while True:
                yield from items
        random.seed(99)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            UDAIICEpocher(model, pw, opt, cyc(lab), cyc(unl), kl, torch.nn.MSELoss(), lw, num_batches=3, cur_epoch=0, device=DEV,
                          feature_position=FEATURES, feature_importance=[0.5, 0.25, 0.25], cons_weight=5.0, iic_weight=0.1).run()
        sc = opt.loss_scaler
        assert sc is not None and sc.overflows == expect_overflows, (sc.overflows, sc.scale)
        moved = any(not torch.equal(before[k], v) for k, v in model.state_dict().items() if k.endswith("weight"))
        if expect_overflows:
            assert sc.scale == init / 4 and not moved and opt._steps[0] == 0
            assert any("overflow" in str(w.message) for w in caught)
            assert all(float(m.abs().max()) == 0 for m in opt._m)            # the moments did not move either
        else:
            assert sc.scale == 16384.0 and moved and sc.good == 3


_CFG4_ORACLE = {}


@pytest.mark.parametrize("dtype,mi_precision", [("float32", "fp32"), ("bfloat16", "f16f8")])
def test_cfg4_step_matches_the_oracle(monkeypatch, dtype, mi_precision):
    """BASELINE configs[3] as a whole step, reduced to what the CPU oracle finishes in a minute: EIGHT classes, local MI over the
    7 x 7 grid of half-overlapping patches (49 windows, +-3 displacement on Up_conv2, +-1 on Up_conv3; ref iic_loss.py:152-189) on
    256 x 256 slices with 64 x 64 patches (configs[3] itself: 512 x 512 with 128 x 128 patches -- the same grid at twice the scale;
    its local-MI launches at full size are checked in test_gpu_mi.py::test_patch_local_mi / test_bf16x3_batched_heads_with_
    overlapping_patches / test_joint_checksum_at_full_size, and `python bench.py --config cfg4` times it at LB = UB = 16) -- one
    `udaiic` iteration with LB = UB = 1 against the oracle on the same weights: every meter, the gradients of the logits layer and
    of the head parameters."""
    from oracle import step as OS
    from contrastyou.arch import UNet
    from deepclustering2.loss import KL_div
    from deepclustering2.optim import Adam
    from semi_seg._utils import IICLossWrapper, ProjectorWrapper
    from semi_seg.epocher import UDAIICEpocher
    from itertools import chain
    from miseg_amd import ops as _ops, unet_ops
    import bigstep
    H, LB, UB, NC, PATCH = 256, 1, 1, 8, 64
    inp = bigstep.cfg4_inputs()
    heads, limg, ltgt, uimg = inp["heads"], inp["limg"], inp["ltgt"], inp["uimg"]
    model = UNet(1, NC, compute_dtype=dtype)
    model.load_state_dict(OU.init_state(1, NC, seed=50))
    pw = ProjectorWrapper()
    pw.init_encoder(feature_names=FEATURES, num_clusters=20, num_subheads=5, head_types="linear", normalize=False)
    pw.init_decoder(feature_names=FEATURES, num_clusters=20, num_subheads=5, head_types="linear", normalize=False)
    pw._encoder_projectors["Conv5"].load_state_dict(heads["Conv5"])
    pw._decoder_projectors["Up_conv3"].load_state_dict(heads["Up_conv3"])
    pw._decoder_projectors["Up_conv2"].load_state_dict(heads["Up_conv2"])
    lw = IICLossWrapper(feature_names=FEATURES, paddings=[1, 3], patch_sizes=PATCH)
    model, pw = model.to(DEV), pw.to(DEV)
    opt = Adam(chain(model.parameters(), pw.parameters()), lr=1e-3, weight_decay=1e-5)

    def loader(img, tgt):
        B = img.shape[0]
        yield [[[img, tgt], [img.clone(), tgt.clone()]], [f"patient{j:03d}_00_{j}" for j in range(B)], ["0"] * B, [f"patient{j:03d}_00" for j in range(B)]]

    grabbed, real = [], unet_ops.adam_step

    def spy(param, grad, *a, **k):
        grabbed.append(grad.detach().clone())
        return real(param, grad, *a, **k)
    monkeypatch.setattr(unet_ops, "adam_step", spy)
    _ops.set_mi_precision(mi_precision)
    try:
        random.seed(1357)
        res = UDAIICEpocher(model, pw, opt, loader(limg, ltgt), loader(uimg, torch.zeros(UB, 1, H, H, dtype=torch.long)), KL_div(verbose=False),
                            torch.nn.MSELoss(), lw, num_batches=1, cur_epoch=0, device=DEV, feature_position=FEATURES,
                            feature_importance=[0.5, 0.25, 0.25], cons_weight=5.0, iic_weight=0.1).run()
    finally:
        _ops.set_mi_precision("fp32")
    # the oracle steps (~1 min of CPU each, fp32 and bf16-emulated): cached in tests/golden/bigstep_oracle.npz (bigstep.py), recomputed
    # if oracle/ or the input generator changed since
    exact = dtype == "float32"
    sc, grads = bigstep.result("cfg4", False)
    if not exact:        # bf16 storage: the oracle with the same rounding points
        sc = bigstep.result("cfg4", True)[0]
    _dump(f"cfg4_scalars_{dtype}_{mi_precision}", {k: (abs(res[k]["mean"] - sc[k]) / (abs(sc[k]) + 1e-30), res[k]["mean"], sc[k])
                                                    for k in ("sup_loss", "uda", "mi", "reg_loss")})
    np.testing.assert_allclose(res["sup_loss"]["mean"], sc["sup_loss"], rtol=2e-5 if exact else 5e-3)
    np.testing.assert_allclose(res["uda"]["mean"], sc["uda"], rtol=2e-4 if exact else 2e-2)
    if exact:
        np.testing.assert_allclose(res["mi"]["mean"], sc["mi"], rtol=2e-3, atol=2e-6)
        for f in FEATURES:
            np.testing.assert_allclose(res["individual_mis"][f], sc[f"mi/{f}"], rtol=2e-3, atol=2e-6)
        np.testing.assert_allclose(res["reg_loss"]["mean"], sc["reg_loss"], rtol=2e-4, atol=1e-7)
    else:
        np.testing.assert_allclose(res["mi"]["mean"], sc["mi"], rtol=5e-2, atol=2e-6)
        for f in FEATURES:
            np.testing.assert_allclose(res["individual_mis"][f], sc[f"mi/{f}"], rtol=5e-2, atol=2e-6)
    flat, fb = grabbed[0].cpu() / float(opt.grad_scale), opt.flat
    pref = {"Conv5": pw._encoder_projectors["Conv5"], "Up_conv3": pw._decoder_projectors["Up_conv3"], "Up_conv2": pw._decoder_projectors["Up_conv2"]}
    worst = {}
    for name, g_ref in grads.items():
        if "/" in name:
            f, k = name.split("/", 1)
            p = dict(pref[f].named_parameters())[k]
        elif name.startswith("DeConv_1x1"):
            p = dict(model.named_parameters())[name]
        else:
            continue
        off = fb.offset_of(p)
        mine, ref = flat[off:off + p.numel()].view(p.shape).double(), g_ref.double()
        if float(ref.abs().max()) > 1e-7:
            worst[name] = float((mine - ref).norm() / ref.norm())
    _dump(f"cfg4_{dtype}_{mi_precision}", {k: (v, v, 0.0) for k, v in worst.items()})
    dec = {k: v for k, v in worst.items() if not k.startswith("Conv5/")}
    assert len(dec) >= 2 + 2 * 2
    if exact:
        assert max(dec.values()) < 2e-3, {k: v for k, v in dec.items() if v >= 2e-3}
        assert max(v for k, v in dec.items() if k.startswith("DeConv_1x1")) < 1e-5, dec
    else:
        # measured: logits layer 3.4e-3 / 1.3e-3, decoder-tap heads 1.7e-3 .. 1.2e-2
        assert max(v for k, v in dec.items() if k.startswith("DeConv_1x1")) < 1e-2, dec
        assert max(dec.values()) < 4e-2, {k: v for k, v in dec.items() if v >= 4e-2}
