"""The reference's entry command (ref semi_seg/main.py:1-45) end to end on the GPU: `python semi_seg/main.py
Trainer.name=... key=value ...` with the shipped YAML, for every trainer of the zoo -- two tiny epochs of training,
the per-epoch val/test evaluation, the checkpoint and the storage CSV.  Synthetic ACDC-shaped data (the PNG pipeline is
outside the hot-path scope)."""
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mi-based-regularized-semi-supervised-segmentation_amd")


@pytest.mark.parametrize("name,dtype", [("partial", "bfloat16"), ("uda", "bfloat16"), ("iic", "bfloat16"), ("udaiic", "bfloat16"),
                                        ("udaiic", "float16")])
def test_main_cli_runs_every_trainer(name, dtype):
    save = f"pytest_cli_{name}_{dtype}"
    run_dir = os.path.join(PKG, "semi_seg", "runs", save)
    shutil.rmtree(run_dir, ignore_errors=True)
    try:
        res = subprocess.run(
            [sys.executable, "semi_seg/main.py", f"Trainer.name={name}", f"Trainer.save_dir={save}", "Trainer.device=cuda",
             "Trainer.max_epoch=2", "Trainer.num_batches=3", "Data.size=64", "LabeledData.batch_size=2",
             "UnlabeledData.batch_size=2", f"Arch.compute_dtype={dtype}"],
            cwd=PKG, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        files = set(os.listdir(run_dir))
        assert {"config.yaml", "last.pth"} <= files, files
    finally:
        shutil.rmtree(run_dir, ignore_errors=True)


def test_main_cli_non_default_iic_configuration():
    """Everything the reference's YAML lets a user change on the IIC side, at once: four taps at four scales (encoder + three decoder
    blocks), mlp heads with normalisation, 10 clusters x 3 sub-heads (off the K = 20 fast path: generic fp32-MFMA local-MI kernels),
    paddings 2 / 1 / 3, 32-pixel patches (overlapping windows), the `kl` consistency criterion -- two tiny epochs must run and
    produce finite meters (ref config/semi.yaml:40-63, semi_seg/trainer.py:137-160)."""
    save = "pytest_cli_variants"
    run_dir = os.path.join(PKG, "semi_seg", "runs", save)
    shutil.rmtree(run_dir, ignore_errors=True)
    try:
        res = subprocess.run(
            [sys.executable, "semi_seg/main.py", "Trainer.name=udaiic", f"Trainer.save_dir={save}", "Trainer.device=cuda",
             "Trainer.max_epoch=2", "Trainer.num_batches=2", "Data.size=64", "LabeledData.batch_size=2", "UnlabeledData.batch_size=3",
             "Arch.compute_dtype=bfloat16", "Trainer.feature_names=[Conv5,Up_conv4,Up_conv3,Up_conv2]",
             "Trainer.feature_importance=[1,0.5,0.5,0.25]", "UDARegCriterion.name=kl",
             "IICRegParameters.EncoderParams.head_types=mlp", "IICRegParameters.EncoderParams.normalize=true",
             "IICRegParameters.EncoderParams.num_clusters=10", "IICRegParameters.EncoderParams.num_subheads=3",
             "IICRegParameters.DecoderParams.head_types=mlp", "IICRegParameters.DecoderParams.normalize=true",
             "IICRegParameters.DecoderParams.num_clusters=10", "IICRegParameters.DecoderParams.num_subheads=3",
             "IICRegParameters.LossParams.paddings=[2,1,3]", "IICRegParameters.LossParams.patch_sizes=32"],
            cwd=PKG, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        rows = open(os.path.join(run_dir, "storage.csv")).read().splitlines()
        assert len(rows) == 3
        header = rows[0].split(",")
        for row in rows[1:]:
            vals = dict(zip(header, row.split(",")))
            for k, v in vals.items():
                if k.startswith("tra_") and v not in ("",):
                    assert float(v) == float(v) and abs(float(v)) < 1e6, (k, v)      # finite
            assert any(k.startswith("tra_individual_mis") or "mi" in k for k in vals)
    finally:
        shutil.rmtree(run_dir, ignore_errors=True)


@pytest.mark.parametrize("extra,expect", [([], "f16f8"), (["Arch.mi_precision=bf16x3"], "bf16x3"), (["Arch.compute_dtype=float32"], "fp32")])
def test_main_entry_point_selects_the_benchmarked_arithmetic(monkeypatch, extra, expect):
    """`python semi_seg/main.py Trainer.name=udaiic Arch.compute_dtype=bfloat16` -- the reference's surface (ref semi_seg/main.py:19-44,
    config/semi.yaml:3-5) -- must run the arithmetic bench.py measures: the f16 + fp8 local-MI split by default for 16-bit storage
    (`Arch.mi_precision` overrides, float32 storage stays exact fp32), no environment variable involved.  The local-MI backward
    launches of the run are watched at the C boundary (the pad-3 tap of K = 20 x 5 sub-heads with precision 3 IS the fp8 kernel,
    csrc/mi_local.hip dispatch), and the later iterations come from the launch tape."""
    sys.path.insert(0, PKG)
    monkeypatch.delenv("MISEG_MI_PRECISION", raising=False)
    import semi_seg.main as M
    from miseg_amd import ops
    seen = []
    real = ops.call

    def spy(name, *a, **k):
        if name == "miseg_iic_local_bwd_heads":
            seen.append((int(a[7]), int(a[14])))          # (pad, precision)
        return real(name, *a, **k)
    monkeypatch.setattr(ops, "call", spy)
    save = f"pytest_cli_prec_{expect}"
    run_dir = os.path.join(PKG, "semi_seg", "runs", save)
    shutil.rmtree(run_dir, ignore_errors=True)
    try:
        argv = ["Trainer.name=udaiic", f"Trainer.save_dir={save}", "Trainer.device=cuda", "Trainer.max_epoch=2", "Trainer.num_batches=6",
                "Data.name=synthetic", "Data.size=64", "LabeledData.batch_size=2", "UnlabeledData.batch_size=2"] + \
            ([] if any(e.startswith("Arch.compute_dtype") for e in extra) else ["Arch.compute_dtype=bfloat16"]) + extra
        trainer = M.main(argv)
        assert ops.mi_precision_name() == expect
        assert seen and all(p == ops.MI_PRECISIONS[expect] for _, p in seen), seen
        assert any(pad == 3 for pad, _ in seen)
        tape = trainer._optimizer._miseg_step_ctx["tape"]
        assert tape is not None and tape.disabled is None and tape.replays >= 6, (tape and tape.disabled, tape and tape.replays)
        assert len(seen) == 2 * 4, seen                   # two decoder taps x (3 eager + 1 recorded) iterations: the rest were replayed
    finally:
        shutil.rmtree(run_dir, ignore_errors=True)
        ops.set_mi_precision("fp32")


def test_trainer_inference_dumps_pngs_and_reports_hausdorff(tmp_path):
    """SemiTrainer.inference (ref semi_seg/trainer.py:109-124 -> InferenceEpocher, epocher.py:76-107)."""
    sys.path.insert(0, PKG)
    import torch
    from contrastyou.arch import UNet
    from deepclustering2.loss import KL_div
    from semi_seg.synthetic import SyntheticEval, SyntheticPairs
    from semi_seg.trainer import trainer_zoos
    cfg = {"Optim": {"name": "Adam", "lr": 1e-4, "weight_decay": 1e-5},
           "Trainer": {"feature_names": ["Conv5", "Up_conv3", "Up_conv2"], "feature_importance": [1, 0.5, 0.5], "max_epoch": 1}}
    tr = trainer_zoos["partial"](
        model=UNet(input_dim=1, num_classes=4), labeled_loader=iter(SyntheticPairs(2, 64, 4, seed=0)),
        unlabeled_loader=iter(SyntheticPairs(2, 64, 4, seed=1)), val_loader=SyntheticEval(1, 2, 64, 4, seed=2),
        test_loader=SyntheticEval(2, 3, 64, 4, seed=3), sup_criterion=KL_div(), configuration=cfg, save_dir=str(tmp_path / "run"),
        max_epoch=1, num_batches=2, device="cuda")
    tr.init()
    tr.start_training()
    result, score = tr.inference()
    assert 0.0 <= score <= 1.0 and "dice" in result and "hd" in result
    for sub in ("img", "gt", "pred"):
        assert len(os.listdir(tmp_path / "run" / sub)) == 6
    from PIL import Image
    import numpy as np
    name = sorted(os.listdir(tmp_path / "run" / "pred"))[0]
    assert np.array(Image.open(tmp_path / "run" / "pred" / name)).max() <= 3


@pytest.mark.parametrize("name", ["partial", "uda", "iic", "udaiic"])
def test_run_directory_matches_the_reference_trainers(golden, tmp_path, name):
    """SURVEY 8(f-3) after training: two tiny epochs of this repo's trainer write the files, the `last.pth` key tree (Adam state per
    parameter, scheduler fields, `_storage` as pickled HistoricalContainer objects, `_buffers`), the `storage.csv` header / index
    and the `config.yaml` keys the REFERENCE trainer of the same name wrote for the same configuration
    (tests/golden/trainer_io.npz, make_golden.py::gen_trainer_io)."""
    sys.path.insert(0, PKG)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import synth
    from test_cpu_host import _build_trainer
    g = golden("trainer_io")
    tr = _build_trainer(name, tmp_path / "run", device="cuda")
    tr.start_training()
    run = tmp_path / "run"
    assert sorted(f for f in os.listdir(run) if not f.startswith("tensorboard")) == [str(x) for x in g[f"{name}/files"]]
    ck = torch.load(run / "last.pth", map_location="cpu", weights_only=False)
    mine, ref = sorted(synth.tree_lines(ck)), [str(x) for x in g[f"{name}/tree_last_pth"]]
    assert mine == ref, (sorted(set(mine) - set(ref))[:12], sorted(set(ref) - set(mine))[:12])
    rows = open(run / "storage.csv").read().splitlines()
    assert rows[0].split(",") == [str(x) for x in g[f"{name}/csv_header"]]
    assert [r.split(",")[0] for r in rows[1:]] == [str(x) for x in g[f"{name}/csv_index"]]
    assert [f"{k}={type(v).__name__}" for k, v in ck["_buffers"].items()] == [str(x) for x in g[f"{name}/buffers"]]


def test_bench_line_contract():
    """`python bench.py` prints ONE JSON line with the driver's fields (metric / value / unit / n_gpus / steps / warmup / ms_per_step /
    higher_is_better / scaling / vs_baseline / dtype / data / config.workload), the `roofline` object of the dominant kernel measured
    with HIP events inside the timed region, and -- through `--gpus 1 --spawn`-style self-launch -- the rank count the process group saw."""
    import json
    res = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "rccl_ranks", "backend"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["dtype"] == "bf16" and d["scaling"] == "weak"
    assert d["unit"] == "images/s" and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 32 * 3 / (d["ms_per_step"] * 3e-3)) < 0.02 * d["value"]            # value = images of the timed steps / time
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_ms", "ceiling_frac", "microbench_sustained_TFLOPs"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["kernel"].startswith("iic_local_bwd") and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 0.5 and r["avg_ms"] < d["ms_per_step"]
