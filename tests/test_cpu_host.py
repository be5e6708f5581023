"""CPU (no GPU): the C-ABI library builds/loads and exports every declared symbol; host-side logic of the
reference-compatible surface (config parser, meters, scheduler, flip decisions, patch windows, checkpoints);
the product refuses CPU tensors instead of falling back."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

import synth
from oracle import iic as OI
from oracle import losses as OL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = torch.from_numpy


def test_cabi_library_exports_every_declared_symbol():
    from miseg_amd import _cabi
    names = _cabi.declared_symbols()
    assert len(names) >= 30 and "miseg_iic_local_joint_fwd" in names and "miseg_conv3x3_fwd" in names
    lib = _cabi.lib()                      # loads without a GPU; raises if the .so is missing
    for n in names:
        assert hasattr(lib, n), n
    assert lib.miseg_version() >= 100
    # nm view: every declared symbol is a defined text symbol of the shared object
    out = subprocess.check_output(["nm", "-D", "--defined-only", _cabi.LIB_PATH]).decode()
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert set(names) <= exported, set(names) - exported


def test_host_side_argument_validation_without_gpu():
    """Entry points validate shapes on the host before any launch (no compute call is made here)."""
    from miseg_amd import _cabi
    assert _cabi.query("miseg_iic_local_joint_ws_bytes", 16, 20, 256, 256, 3, 1) > 0
    with pytest.raises(_cabi.MisegError):
        _cabi.query("miseg_iic_local_joint_ws_bytes", 16, 20, 256, 256, 3, 0)        # P must be > 0
    with pytest.raises(_cabi.MisegError, match="null pointer"):
        _cabi.call("miseg_iic_local_joint_fwd", None, None, None, None, 1, 1, 1, 1, 0, None, 1, None, None, 0, 0)


def test_product_refuses_cpu_tensors():
    from miseg_amd import _cabi, ops
    x = torch.rand(2, 4, 8, 8).softmax(1).requires_grad_(True)
    with pytest.raises(_cabi.MisegError, match="GPU only"):
        ops.local_mi_losses(x, x, 1, [(0, 8, 0, 8)])
    from contrastyou.losses.iic_loss import IIDSegmentationLoss
    with pytest.raises(_cabi.MisegError):
        IIDSegmentationLoss(padding=1)(x, x)
    from contrastyou.arch import UNet
    with pytest.raises(_cabi.MisegError):
        UNet(1, 4)(torch.rand(1, 1, 32, 32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mi-based-regularized-semi-supervised-segmentation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, os.path.join(dirpath, f)


@pytest.mark.parametrize("h,patch", [(100, 32), (64, 1024), (512, 128), (224, 1024), (48, 16), (33, 16)])
def test_patch_windows_match_reference_and_colouring_is_disjoint(golden, h, patch):
    from contrastyou.losses.iic_loss import _windows, patch_generator
    from miseg_amd.ops import colour_windows
    wins = _windows(h, h, (patch, patch), (patch // 2, patch // 2))
    np.testing.assert_array_equal(np.asarray(wins, dtype=np.int64), golden("iic")[f"patchgeom_h{h}_ps{patch}/windows"])
    fm = torch.arange(h * h, dtype=torch.float32).view(1, 1, h, h)
    for (h0, h1, w0, w1), crop in zip(wins, patch_generator(fm, (patch, patch), (patch // 2, patch // 2))):
        assert crop.shape[2:] == (h1 - h0, w1 - w0) and float(crop[0, 0, 0, 0]) == h0 * h + w0
    groups = colour_windows(wins)
    assert sorted(i for g in groups for i in g) == list(range(len(wins)))
    for g in groups:
        for a in g:
            for b in g:
                if a < b:
                    (a0, a1, a2, a3), (b0, b1, b2, b3) = wins[a], wins[b]
                    assert a1 <= b0 or b1 <= a0 or a3 <= b2 or b3 <= a2


@pytest.mark.parametrize("seed", [0, 123, 9999999, 4242])
def test_flip_decisions_replay(golden, seed):
    from deepclustering2.augment.tensor_augment import TensorRandomFlip
    from deepclustering2.decorator import FixRandomSeed
    flipper = TensorRandomFlip(axis=[1, 2], threshold=0.8)
    random.seed(555)
    before = random.random()
    random.seed(555)
    with FixRandomSeed(seed):
        dec = flipper.decisions(4)
    assert random.random() == before
    np.testing.assert_array_equal(np.asarray(dec), golden("losses")[f"flip/seed{seed}/decisions"])
    x = torch.arange(4 * 2 * 3 * 5, dtype=torch.float32).view(4, 2, 3, 5)
    with FixRandomSeed(seed):
        legacy = torch.stack([flipper(s) for s in x])          # per-sample call path kept for API compatibility
    np.testing.assert_array_equal(legacy.numpy(), golden("losses")[f"flip/seed{seed}/out"])
    from miseg_amd.ops import flips_to_tensor
    assert flips_to_tensor(dec, "cpu").tolist() == [int(a) | (int(b) << 1) for a, b in dec]


def test_meters_and_schedule_match_reference(golden):
    from deepclustering2.meters2 import AverageValueMeter, MeterInterface, MultipleAverageValueMeter, UniversalDice
    from deepclustering2.schedulers import GradualWarmupScheduler
    g = golden("meters_sched")
    meter = UniversalDice(4, report_axises=[1, 2, 3])
    for it in range(3):
        meter.add(T(synth.integers(f"dice/pred{it}", (4, 12, 12), 4)), T(synth.integers(f"dice/target{it}", (4, 12, 12), 4)),
                  group_name=[str(s) for s in g["dice/groups"][it]])
    summ = meter.summary()
    assert list(summ.keys()) == [str(k) for k in g["dice/keys"]]
    np.testing.assert_allclose(list(summ.values()), g["dice/values"], rtol=1e-6)
    avg = AverageValueMeter()
    for v, m in zip(g["avg/seq"], g["avg/means"]):
        avg.add(float(v))
        np.testing.assert_allclose(avg.summary()["mean"], m, rtol=1e-12)
    mi = MeterInterface()
    mi.register_meter("a", AverageValueMeter())
    mi.register_meter("b", MultipleAverageValueMeter())
    mi["a"].add(1.0), mi["b"].add(x=2.0, y=3.0)
    assert dict(mi.tracking_status()["b"]) == {"x": 2.0, "y": 3.0}
    for max_epoch, warm, mult in ((100, 10, 400), (30, 5, 300)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=1e-7)
        cos = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=max_epoch - warm, eta_min=1e-7)
        sched = GradualWarmupScheduler(opt, mult, total_epoch=warm, after_scheduler=cos)
        lrs = []
        for _ in range(max_epoch):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
        np.testing.assert_allclose(lrs, g[f"sched_e{max_epoch}_w{warm}_m{mult}/lrs"], rtol=1e-9)


def test_config_manager_cli_overrides(tmp_path):
    from deepclustering2.configparser import ConfigManger
    cfg_path = os.path.join(ROOT, "mi-based-regularized-semi-supervised-segmentation_amd", "config", "semi.yaml")
    cm = ConfigManger(cfg_path, verbose=False, argv=["Trainer.name=udaiic", "Optim.lr=0.001", "IICRegParameters.LossParams.paddings=[1,3]",
                                                      "Trainer.save_dir=x/y"])
    c = cm.config
    assert c["Trainer"]["name"] == "udaiic" and c["Optim"]["lr"] == 0.001 and c["Trainer"]["save_dir"] == "x/y"
    assert c["IICRegParameters"]["LossParams"]["paddings"] == [1, 3]
    assert c["Trainer"]["feature_names"] == ["Conv5", "Up_conv3", "Up_conv2"] and c["Arch"] == {"input_dim": 1, "num_classes": 4}
    assert cm.default_config["Trainer"]["name"] == "partial"


def test_trainer_wiring_and_checkpoint_roundtrip(tmp_path):
    """udaiic trainer builds the reference's object graph on CPU (no kernels run) and checkpoints round-trip."""
    import yaml
    from contrastyou.arch import UNet
    from deepclustering2.loss import KL_div
    from semi_seg.synthetic import SyntheticEval, SyntheticPairs
    from semi_seg.trainer import trainer_zoos
    cfg = yaml.safe_load(open(os.path.join(ROOT, "mi-based-regularized-semi-supervised-segmentation_amd", "config", "semi.yaml")))
    cfg["Trainer"].update(name="udaiic", save_dir=str(tmp_path / "run"), device="cpu", max_epoch=2, num_batches=1)
    tcfg = {k: v for k, v in cfg["Trainer"].items() if k != "name"}
    tr = trainer_zoos["udaiic"](model=UNet(**cfg["Arch"]), labeled_loader=iter(SyntheticPairs(2, 32)), unlabeled_loader=iter(SyntheticPairs(2, 32)),
                                val_loader=SyntheticEval(1, 2, 32), test_loader=SyntheticEval(1, 2, 32), sup_criterion=KL_div(),
                                configuration=cfg, **tcfg)
    tr.init()
    assert tr.feature_positions == ["Conv5", "Up_conv3", "Up_conv2"]
    np.testing.assert_allclose(tr._feature_importance, [0.5, 0.25, 0.25])
    assert tr._iic_weight == 0.1 and tr._uda_weight == 5.0 and tr._reg_weight == 1.0
    n_model = sum(p.numel() for p in tr._model.parameters())
    n_proj = sum(p.numel() for p in tr._projector_wrappers.parameters())
    assert (n_model, n_proj) == (2160180, 30700)
    assert [type(c).__name__ for c in tr._IIDSegWrapper] == ["IIDLoss", "IIDSegmentationSmallPathLoss", "IIDSegmentationSmallPathLoss"]
    assert os.path.exists(tmp_path / "run" / "config.yaml")
    tr._cur_epoch, tr._best_score = 3, 0.5
    tr._save_to("last.pth")
    sd = torch.load(tmp_path / "run" / "last.pth", weights_only=False)
    assert {"_model", "_optimizer", "_scheduler", "_projector_wrappers", "_IIDSegWrapper", "_sup_criterion", "_reg_criterion",
            "_storage", "_buffers"} <= set(sd)
    assert sd["_buffers"] == {"_best_score": 0.5, "_start_epoch": 0, "_cur_epoch": 3}
    with torch.no_grad():
        next(tr._model.parameters()).add_(1.0)
    tr.load_state_dict_from_path(str(tmp_path / "run"), strict=True)
    assert tr._start_epoch == 4
    torch.testing.assert_close(tr._model.state_dict()["Conv1.conv.0.weight"], sd["_model"]["Conv1.conv.0.weight"])


def test_adjacent_head_params_are_stacked_views_of_the_flat_buffers():
    """gradslot.register_adjacent: sub-head parameters sit back to back in the flat buffers, stacked_param is a view, the
    gradient arrives per parameter, and the optimiser state keeps the torch parameter order."""
    import torch
    from miseg_amd.flat import FlatBuffers
    from miseg_amd.gradslot import register_adjacent, stacked_param
    torch.manual_seed(0)
    heads = torch.nn.ModuleList([torch.nn.Linear(8, 4) for _ in range(3)])
    other = torch.nn.Linear(4, 4)
    ws, bs = [h.weight for h in heads], [h.bias for h in heads]
    register_adjacent(ws)
    register_adjacent(bs)
    ref_w = torch.stack([w.detach().clone() for w in ws])
    assert stacked_param(ws).data_ptr() != ws[0].data_ptr()          # not adjacent yet: plain stack
    params = [heads[0].weight, heads[0].bias, other.weight, heads[1].weight, heads[1].bias, other.bias, heads[2].weight, heads[2].bias]
    fb = FlatBuffers(params)
    fb.build()
    assert sorted(fb.offsets) == fb.offsets
    sw, sb = stacked_param(ws), stacked_param(bs)
    assert sw.data_ptr() == ws[0].data_ptr() and sw.shape == (3, 4, 8) and sb.shape == (3, 4)
    assert torch.equal(sw, ref_w)
    coef = torch.randn(3, 4, 8)
    ((sw * coef).sum() + (sb * 2).sum()).backward()
    fb.collect()
    for i in range(3):
        assert torch.equal(ws[i].grad, coef[i]) and torch.equal(bs[i].grad, torch.full((4,), 2.0))
        o = fb.offset_of(ws[i])
        assert torch.equal(fb.flat_grad[o:o + 32].view(4, 8), coef[i])


def test_surface_meter_hausdorff_against_brute_force():
    """SurfaceMeter (MedPy 0.4.0 __surface_distances restated on scipy) vs an O(n^2) Hausdorff of the border pixels."""
    import numpy as np
    import torch
    from scipy.ndimage import binary_erosion, generate_binary_structure
    from deepclustering2.meters2 import SurfaceMeter
    rng = np.random.default_rng(0)

    def border(m):
        return m ^ binary_erosion(m, structure=generate_binary_structure(2, 1), iterations=1)

    def brute(a, b):
        pa, pb = np.argwhere(border(a)), np.argwhere(border(b))
        d = np.sqrt(((pa[:, None, :] - pb[None, :, :]) ** 2).sum(-1))
        return max(d.min(1).max(), d.min(0).max())

    pred = torch.zeros(3, 24, 24, dtype=torch.int64)
    tgt = torch.zeros(3, 24, 24, dtype=torch.int64)
    for b in range(3):
        for c in (1, 2):
            y, x = rng.integers(2, 12, 2)
            pred[b, y:y + 6 + c, x:x + 5] = c
            tgt[b, y + 1:y + 8, x + c:x + 7] = c
    m = SurfaceMeter(C=3, report_axises=[1, 2])
    m.add(pred, tgt)
    want = np.array([[brute(pred[b].numpy() == c, tgt[b].numpy() == c) for c in (1, 2)] for b in range(3)])
    got = m.summary()
    assert abs(got["HD1"] - want[:, 0].mean()) < 1e-9 and abs(got["HD2"] - want[:, 1].mean()) < 1e-9
    import pytest
    with pytest.raises(RuntimeError):                      # an absent class aborts the batch, as MedPy does
        m.add(torch.zeros(1, 8, 8, dtype=torch.int64), tgt[:1, :8, :8])
    assert m._n == 1


def test_deferred_readback_ticket_order_and_flush():
    """Host logic of the one-iteration-late read-back (semi_seg/epocher.py `_Pending.post` / `wait`, `TrainEpocher._after_step`):
    iteration i is recorded when iteration i+1 posts, the last one by `_flush_records`, every iteration exactly once and in
    order, with the Dice counts of ITS iteration; MISEG_DEFER_FETCH=0 semantics (`_DEFER_FETCH = False`) record immediately.
    CPU tensors take the synchronous branch of `post`, so this runs without a GPU."""
    from semi_seg.epocher import TrainEpocher, _Pending

    class Probe(TrainEpocher):
        def __init__(self, defer):   # noqa: the loop helpers only need these three attributes
            self._pending, self._inflight, self._DEFER_FETCH, self.seen = _Pending(), None, defer, []

        def _record(self, host, inter, union, label_group):
            self.seen.append((host["sup_loss"], int(inter.sum()), int(union.sum()), label_group))

    for defer in (True, False):
        ep = Probe(defer)
        for i in range(4):
            ep._pending.put("sup_loss", torch.tensor(float(i)))
            ep._pending.put("reg_loss", torch.tensor(0.5))
            ep._after_step(torch.full((2, 3), i), torch.full((2, 3), 10 * i), f"g{i}")
            assert len(ep.seen) == (i if defer else i + 1)
        ep._flush_records()
        ep._flush_records()   # idempotent
        assert ep.seen == [(float(i), 6 * i, 60 * i, f"g{i}") for i in range(4)]
    # a failed deferred assertion surfaces from wait(), i.e. one iteration late but never lost
    from miseg_amd import checks
    pend = _Pending()
    with checks.deferred(pend.checks):
        checks.require_zero(torch.tensor(3), AssertionError, "three bad pixels")
    ticket = pend.post()
    with pytest.raises(AssertionError, match="three bad pixels"):
        _Pending.wait(ticket)
    # the NaN test of a loss is recorded lazily inside a deferred block (its kernels run when the flags are gathered) and raises
    # the reference's RuntimeError from wait(); outside a block it raises in line
    pend = _Pending()
    with checks.deferred(pend.checks):
        checks.raise_if_nan(torch.tensor([0.25, float("nan")]), "a patch loss is nan")
        checks.raise_if_nan(torch.tensor([0.25, 1.0]), "never")
    assert callable(pend.checks[0][0])
    with pytest.raises(RuntimeError, match="a patch loss is nan"):
        _Pending.wait(pend.post())
    with pytest.raises(RuntimeError, match="inline"):
        checks.raise_if_nan(torch.tensor(float("nan")), "inline")


def test_pending_precompute_gives_the_guard_flags_and_keeps_the_report_for_the_post():
    """`_Pending.precompute` (called before the optimiser launch): evaluates values and check flags once, returns the flag part
    (the device-side guard of the Adam launch) and leaves the vector for the iteration's post / fetch -- same values, same
    exceptions, nothing evaluated twice and nothing left over for the next iteration."""
    from miseg_amd import checks
    from miseg_amd.lazy import LinearLoss
    from semi_seg.epocher import _Pending
    pend = _Pending()
    a, b = torch.tensor(2.0), torch.tensor([1.0, 3.0])
    pend.put("sup_loss", LinearLoss.of(a) * 0.5)
    pend.put("mi", -LinearLoss.mean(b))
    assert pend.precompute() is None or pend.precompute().numel() == 0            # no checks recorded: nothing to guard with
    assert pend.fetch() == {"sup_loss": 1.0, "mi": -2.0} and pend.fetch() == {}
    with checks.deferred(pend.checks):
        checks.require_zero(torch.tensor(0, dtype=torch.int32), AssertionError, "simplex")
        checks.raise_if_nan(torch.tensor([0.5, float("nan"), float("nan")]), "a patch loss is nan")
    pend.put("sup_loss", a)
    flags = pend.precompute()
    assert flags.tolist() == [0.0, 2.0]                                           # integer counter cast, number of NaNs
    assert not pend.checks and not pend._vals                                     # moved into the kept report
    assert pend.precompute().tolist() == [0.0, 2.0]                               # idempotent until consumed
    with pytest.raises(RuntimeError, match="a patch loss is nan"):
        _Pending.wait(pend.post())
    assert pend.precompute() is None


def _build_trainer(name, save_dir, device="cpu", size=32):
    import yaml
    from contrastyou.arch import UNet
    from deepclustering2.loss import KL_div
    from semi_seg.synthetic import SyntheticEval, SyntheticPairs
    from semi_seg.trainer import trainer_zoos
    cfg = yaml.safe_load(open(os.path.join(ROOT, "mi-based-regularized-semi-supervised-segmentation_amd", "config", "semi.yaml")))
    cfg["Trainer"].update(name=name, device=device, max_epoch=2, num_batches=1, save_dir=str(save_dir))
    cfg["Scheduler"]["warmup_max"] = 1
    cfg["Trainer"].pop("name")               # as semi_seg/main.py does before it hands the config to the trainer
    tcfg = dict(cfg["Trainer"])
    dev = None if device == "cpu" else device
    tr = trainer_zoos[name](model=UNet(**cfg["Arch"]), labeled_loader=iter(SyntheticPairs(1, size, device=dev)),
                            unlabeled_loader=iter(SyntheticPairs(1, size, device=dev)), val_loader=SyntheticEval(2, 2, size),
                            test_loader=SyntheticEval(2, 2, size), sup_criterion=KL_div(verbose=False),
                            configuration={**cfg, "GITHASH": "none"}, **tcfg)
    tr.init()
    return tr


@pytest.mark.parametrize("name", ["partial", "uda", "iic", "udaiic"])
def test_checkpoint_key_tree_after_init_is_the_reference_trainers(golden, tmp_path, name):
    """the nested key tree of ``trainer.state_dict()`` -- attribute names, parameter / buffer keys, shapes, dtypes,
    optimiser param_group keys, scheduler fields, criterion entries, `_buffers` -- line for line what the REFERENCE trainer of the
    same name produces after ``init()`` (tests/golden/trainer_io.npz, written by make_golden.py::gen_trainer_io)."""
    import synth
    g = golden("trainer_io")
    tr = _build_trainer(name, tmp_path / "run")
    mine = sorted(synth.tree_lines(tr.state_dict()))
    ref = [str(x) for x in g[f"{name}/tree_after_init"]]
    assert mine == ref, (sorted(set(mine) - set(ref))[:12], sorted(set(ref) - set(mine))[:12])
    assert sorted(f"{k}.{kk}" if isinstance(v, dict) else k for k, v in __import__("yaml").safe_load(open(tmp_path / "run" / "config.yaml")).items()
                  for kk in (v if isinstance(v, dict) else [None])) == [str(x) for x in g[f"{name}/config_yaml_keys"]]


def test_reference_layout_checkpoint_loads_strictly(golden, tmp_path):
    """A checkpoint with the layout the reference's ``last.pth`` has after two epochs (every tensor of the fixture's key tree,
    Adam state per parameter, the pickled ``defaultdict(HistoricalContainer)`` history) loads with strict=True and resumes."""
    from collections import OrderedDict, defaultdict
    from deepclustering2.meters2.historicalContainer import HistoricalContainer
    import re
    g = golden("trainer_io")
    tr = _build_trainer("udaiic", tmp_path / "run")
    ck = tr.state_dict()                     # right attribute set and hyper-parameter leaves; now give it the trained-run shape
    lines = [str(x) for x in g["udaiic/tree_last_pth"]]
    state = {}
    for ln in lines:
        m = re.match(r"_optimizer/state/(\d+)/(\w+) :: tensor float32 \[(.*)\]", ln)
        if m:
            shape = [int(v) for v in m.group(3).split(",")] if m.group(3) else []
            state.setdefault(int(m.group(1)), {})[m.group(2)] = torch.full(shape, 2.0 if m.group(2) == "step" else 0.25)
    assert len(state) == 98
    ck["_optimizer"]["state"] = state
    hist = defaultdict(HistoricalContainer)
    for ln in lines:
        if ln.startswith("_storage/"):
            name = ln.split(" :: ")[0].split("/", 1)[1]
            for e in range(2):
                hist[name].add({"mean": float(e)}, e)
    ck["_storage"] = hist
    ck["_buffers"] = OrderedDict(_best_score=0.25, _start_epoch=0, _cur_epoch=1)
    torch.save(ck, tmp_path / "ref_like.pth")
    tr2 = _build_trainer("udaiic", tmp_path / "run2")
    tr2.load_state_dict_from_path(str(tmp_path / "ref_like.pth"), strict=True)
    assert tr2._start_epoch == 2 and tr2._best_score == 0.25
    assert sorted(tr2._storage.meter_names) == sorted(hist) and tr2._storage.summary().shape[0] == 2
    assert tr2._optimizer._pending_state is not None or tr2._optimizer._steps[0] == 2     # applied now (GPU) or when the parameters reach the GPU


def test_lazy_evaluate_isolates_non_finite_values_and_keeps_flags_out_of_the_product():
    """One NaN / inf scalar must not turn every reported scalar (and every deferred-check flag) into NaN: 0 * NaN = NaN in the
    coefficient mat-vec.  Rows that do not use the bad value keep their values; flags pass through untouched."""
    from miseg_amd import lazy
    a, b, c = torch.tensor([1.0, 2.0]), torch.tensor(float("nan")), torch.tensor(float("inf"))
    la, lb, lc = lazy.LinearLoss.mean(a), lazy.LinearLoss.of(b) * 2.0, lazy.LinearLoss.of(c)
    out = lazy.evaluate([la, lb, la + lb, lc, la * 3.0], passthrough=[torch.tensor(0.0), torch.tensor(3.0)])
    assert out[0] == 1.5 and out[4] == 4.5
    assert torch.isnan(out[1]) and torch.isnan(out[2]) and not torch.isfinite(out[3])
    assert out[5] == 0.0 and out[6] == 3.0
    # the epocher's container: a NaN loss raises ITS RuntimeError, not the first registered (passing) check
    from miseg_amd import checks
    from semi_seg.epocher import _Pending
    pend = _Pending()
    with checks.deferred(pend.checks):
        checks.require_zero(torch.tensor(0.0), AssertionError, "simplex")
        checks.raise_if_nan(torch.tensor([float("nan")]), "loss is nan")
        pend.put("loss", lb)
        pend.put("other", la)
    with pytest.raises(RuntimeError, match="loss is nan"):
        pend.fetch()


def test_loss_scaler_policy_and_state():
    """flat.LossScaler: the overflow count of an iteration arrives one iteration late.  An overflow at the current scale halves it; the
    iteration that had already been staged with the old scale overflows too but must not halve again; `growth_interval` clean
    iterations double it; state round-trips through state_dict (the optimiser checkpoints it in the fp16 mode)."""
    from miseg_amd.flat import LossScaler
    sc = LossScaler(1024.0, growth_interval=3, max_scale=4096.0)
    sc.update(5.0, used_scale=1024.0)
    assert sc.scale == 512.0 and sc.overflows == 1
    sc.update(2.0, used_scale=1024.0)            # staged before the first count arrived: counted, not halved again
    assert sc.scale == 512.0 and sc.overflows == 2
    sc.update(float("nan"), used_scale=512.0)
    assert sc.scale == 256.0 and sc.overflows == 3
    for _ in range(3):
        sc.update(0.0, used_scale=256.0)
    assert sc.scale == 512.0 and sc.good == 0
    sd = sc.state_dict()
    other = LossScaler(1.0)
    other.load_state_dict(sd)
    assert (other.scale, other.good, other.overflows) == (sc.scale, sc.good, sc.overflows)
    sc.update(1.0)                               # no scale given (stand-alone use): halves
    assert sc.scale == 256.0


def test_bigstep_oracle_cache_is_current():
    """tests/golden/bigstep_oracle.npz caches four oracle steps that cost minutes of CPU (tests/golden/bigstep.py); the GPU tests use it
    only while the fingerprint of oracle/ + synth.py + bigstep.py stored in it matches.  A mismatch here means: regenerate it
    (`python tests/golden/bigstep.py`), or the GPU suite silently spends those minutes again."""
    import bigstep
    import numpy as np
    assert os.path.exists(bigstep.CACHE), "tests/golden/bigstep_oracle.npz is missing: python tests/golden/bigstep.py"
    z = np.load(bigstep.CACHE, allow_pickle=False)
    assert str(z["fingerprint"]) == bigstep.fingerprint(), "oracle/ or synth.py changed: regenerate with python tests/golden/bigstep.py"
    for key in ("cfg2/fp32", "cfg2/bf16", "cfg4/fp32", "cfg4/bf16"):
        names = [str(n) for n in z[f"{key}/scalar_names"]]
        assert {"sup_loss", "uda", "mi", "reg_loss", "mi/Up_conv2"} <= set(names), (key, names)
        assert len(z[f"{key}/grad_names"]) >= 8
