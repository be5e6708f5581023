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


@pytest.mark.parametrize("name", ["partial", "uda", "iic", "udaiic"])
def test_main_cli_runs_every_trainer(name):
    save = f"pytest_cli_{name}"
    run_dir = os.path.join(PKG, "semi_seg", "runs", save)
    shutil.rmtree(run_dir, ignore_errors=True)
    try:
        res = subprocess.run(
            [sys.executable, "semi_seg/main.py", f"Trainer.name={name}", f"Trainer.save_dir={save}", "Trainer.device=cuda",
             "Trainer.max_epoch=2", "Trainer.num_batches=3", "Data.size=64", "LabeledData.batch_size=2",
             "UnlabeledData.batch_size=2", "Arch.compute_dtype=bfloat16"],
            cwd=PKG, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        files = set(os.listdir(run_dir))
        assert {"config.yaml", "last.pth"} <= files, files
    finally:
        shutil.rmtree(run_dir, ignore_errors=True)
