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
    params = chain(model.parameters(), pw.parameters()) if mode == "udaiic" else model.parameters()
    opt = Adam(params, lr=STEP["lr"], weight_decay=STEP["wd"])

    def loader(tag, B, with_tgt):
        for i in range(NB):
            img = T(synth.uniform(f"step/{mode}/{tag}{i}", (B, 1, H, H)))
            tgt = T(synth.integers(f"step/{mode}/tgt{i}", (B, 1, H, H), 4)) if with_tgt else torch.zeros(B, 1, H, H, dtype=torch.long)
            yield [[[img, tgt], [img.clone(), tgt.clone()]], [f"patient{j:03d}_00_{j}" for j in range(B)], ["0"] * B,
                   [f"patient{j:03d}_00" for j in range(B)]]

    return model, pw, lw, opt, loader("lab", LB, True), loader("unl", UB, False), KL_div(verbose=False)


@pytest.mark.parametrize("mode", ["udaiic", "partial"])
def test_epocher_matches_reference_run(golden, mode):
    from semi_seg.epocher import TrainEpocher, UDAIICEpocher
    g = golden("step")
    model, pw, lw, opt, lab, unl, kl = build(mode)
    fi = [float(v) for v in g[f"{mode}/feature_importance"]]
    random.seed(1234)
    if mode == "udaiic":
        ep = UDAIICEpocher(model, pw, opt, lab, unl, kl, torch.nn.MSELoss(), lw, num_batches=STEP["NB"], cur_epoch=0, device=DEV,
                           feature_position=FEATURES, feature_importance=fi, cons_weight=STEP["cons_weight"], iic_weight=STEP["iic_weight"])
    else:
        ep = TrainEpocher(model, opt, lab, unl, kl, 0, STEP["NB"], 0, DEV, feature_position=FEATURES, feature_importance=fi)
    res = ep.run()
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
    if mode == "udaiic":
        np.testing.assert_allclose(got["uda/mean"], ref["uda/mean"], rtol=0.2)
        for k in ("mi/mean", "individual_mis/Conv5", "individual_mis/Up_conv3", "individual_mis/Up_conv2"):
            np.testing.assert_allclose(got[k], ref[k], rtol=0.2, atol=5e-6)
        assert got["iic_weight/mean"] == ref["iic_weight/mean"] and got["uda_weight/mean"] == ref["uda_weight/mean"]
    # weights after two Adam steps: every entry moved by at most ~lr per step; sign-noise gradients bound the deviation
    for k, v in model.state_dict().items():
        fp = synth.fp_unpack(g, f"{mode}/model_after/{k}")
        synth.check_fingerprint(v.detach().float().cpu().numpy(), fp, f"{mode}/model_after/{k}", rtol=2e-3,
                                atol=2.5 * STEP["lr"] * STEP["NB"])
    moved = float((model.state_dict()["Conv1.conv.0.weight"].cpu() - OU.init_state(1, 4, seed=9)["Conv1.conv.0.weight"]).abs().mean())
    assert 0.2 * STEP["lr"] < moved < 2.5 * STEP["lr"] * STEP["NB"], moved   # the optimiser really stepped


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


@pytest.mark.parametrize("mode", ["udaiic", "partial"])
def test_first_iteration_is_tight(golden, mode):
    """Iteration 1 (identical weights on both sides): every meter of the HIP epocher vs the oracle step, which is
    itself pinned to the reference run by tests/test_oracle_golden.py::test_full_step."""
    from oracle import step as OS
    from oracle import losses as OL
    from semi_seg.epocher import TrainEpocher, UDAIICEpocher
    g = golden("step")
    model, pw, lw, opt, lab, unl, kl = build(mode)
    fi = [float(v) for v in g[f"{mode}/feature_importance"]]
    random.seed(1234)
    if mode == "udaiic":
        ep = UDAIICEpocher(model, pw, opt, lab, unl, kl, torch.nn.MSELoss(), lw, num_batches=1, cur_epoch=0, device=DEV,
                           feature_position=FEATURES, feature_importance=fi, cons_weight=STEP["cons_weight"], iic_weight=STEP["iic_weight"])
    else:
        ep = TrainEpocher(model, opt, lab, unl, kl, 0, 1, 0, DEV, feature_position=FEATURES, feature_importance=fi)
    res = ep.run()
    H, LB, UB = STEP["H"], STEP["LB"], STEP["UB"]
    heads = {"Conv5": OH.init_cluster_head(256, 20, 5, "linear", seed=10), "Up_conv3": OH.init_local_cluster_head(32, 20, 5, "linear", seed=11),
             "Up_conv2": OH.init_local_cluster_head(16, 20, 5, "linear", seed=12)}
    state = OS.StepState(OU.init_state(1, 4, seed=9), heads if mode == "udaiic" else {}, lr=STEP["lr"], weight_decay=STEP["wd"])
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
    if mode == "udaiic":
        np.testing.assert_allclose(res["uda"]["mean"], sc["uda"], rtol=2e-4)
        np.testing.assert_allclose(res["mi"]["mean"], sc["mi"], rtol=2e-3, atol=2e-6)
        for f in FEATURES:
            np.testing.assert_allclose(res["individual_mis"][f], sc[f"mi/{f}"], rtol=2e-3, atol=2e-6)


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
    with pytest.raises((RuntimeError, AssertionError)):
        ep.run()


def test_step_graph_replay_equals_eager_steps():
    """The captured hipGraph of the device half of an iteration (miseg_amd.graph.StepGraph) must reproduce the eager
    iterations: same meters and bit-identical parameters after a mix of eager warm-up, capture and replays.
    Runs in a fresh interpreter: the capture must not follow an eager backward on the default stream in the same process
    (miseg_amd/graph.py docstring), which the other tests of this session have already done."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import random, sys, torch
sys.path[:0] = [%r, %r]
import bench
from miseg_amd import _cabi, ops
_cabi.lib(); ops.set_mi_precision("fp32")
def run(graph):
    torch.manual_seed(0); random.seed(7)
    ep, opt = bench.build_step(torch.device("cuda"), 2, 2, 64, "float32", 0)
    drv = bench.StepDriver(ep)
    if graph:
        ep.enable_step_graph(warmup=2)
    random.seed(11)
    for _ in range(5):   # graph: 2 eager, 1 capture + replay, 2 replays
        drv.step()
    drv.close()
    return opt.flat.flat_param.detach().clone(), dict(ep.meters.tracking_status()), opt._steps[0]
p_graph, m_graph, n_graph = run(True)      # first: nothing has run a backward on the default stream yet
p_eager, m_eager, n_eager = run(False)
assert n_eager == n_graph == 5, (n_eager, n_graph)
assert torch.equal(p_eager, p_graph), float((p_eager - p_graph).abs().max())
assert repr(m_eager) == repr(m_graph), (m_eager, m_graph)   # repr: the unused lr meter is nan in both
print("GRAPH_EQUALS_EAGER")
""" % (root, os.path.join(root, "mi-based-regularized-semi-supervised-segmentation_amd"))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "GRAPH_EQUALS_EAGER" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


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
