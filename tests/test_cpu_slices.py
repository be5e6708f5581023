"""Input pipeline, CPU side: the oracle against the reference-generated golden vectors, the host
planner / job encoder against the oracle (through the numpy interpreter of the job table), the dataset index and split."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import augment as OA  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "augment.npz"))


def _golden_cases():
    for key in GOLD.files:
        if key.endswith("/0/img"):
            name, k, seed, _, _ = key.split("/")
            yield name, int(k), int(seed)


def test_oracle_reproduces_reference_transform_outputs():
    n = 0
    for name, k, seed in _golden_cases():
        out = OA.apply(name, GOLD[f"img{k}"], GOLD[f"gt{k}"], seed)
        views = out if name != "val" else [out]
        for v, (ti, tg) in enumerate(views):
            assert ti.dtype == torch.float32 and tg.dtype == torch.int64
            want = torch.from_numpy(GOLD[f"{name}/{k}/{seed}/{v}/img"]).float().div(255)
            assert torch.equal(ti[0], want), (name, k, seed, v)
            assert np.array_equal(tg[0].numpy(), GOLD[f"{name}/{k}/{seed}/{v}/gt"].astype(np.int64)), (name, k, seed, v)
            n += 1
    assert n >= 20


def _plan_and_interpret(recipe, seed, img, gt):
    from miseg_amd import slices as S
    h, w = img.shape
    plans = S.plan_item(recipe, seed, w, h)
    jobs = S.encode_jobs(plans, [0] * len(plans))
    return OA.run_jobs_numpy(jobs, img[None], gt[None], plans[0].out_h, plans[0].out_w), plans


def test_planner_matches_golden_through_job_table():
    from semi_seg.augment import ACDCStrongTransforms as T
    for name, k, seed in _golden_cases():
        (oi, og), plans = _plan_and_interpret(getattr(T, name), seed, GOLD[f"img{k}"], GOLD[f"gt{k}"])
        for v in range(len(plans)):
            assert np.array_equal(oi[v], GOLD[f"{name}/{k}/{seed}/{v}/img"].astype(np.float32) / np.float32(255)), (name, k, seed, v)
            assert np.array_equal(og[v], GOLD[f"{name}/{k}/{seed}/{v}/gt"].astype(np.int64)), (name, k, seed, v)


@pytest.mark.parametrize("hw", [(224, 224), (256, 256), (230, 301), (257, 224)])
def test_planner_matches_pil_oracle_on_random_slices(hw):
    from semi_seg.augment import ACDCStrongTransforms as T
    rng = np.random.default_rng(hw[0] * 1000 + hw[1])
    img = rng.integers(0, 256, hw, dtype=np.uint8)
    gt = rng.integers(0, 4, hw, dtype=np.uint8)
    for name in ("pretrain", "label", "val", "trainval"):
        for seed in (1, 50000, 31337):
            ref = OA.apply(name, img, gt, seed)
            (oi, og), plans = _plan_and_interpret(getattr(T, name), seed, img, gt)
            for v, (ri, rg) in enumerate(ref if name != "val" else [ref]):
                assert np.array_equal(ri[0].numpy(), oi[v]) and np.array_equal(rg[0].numpy(), og[v]), (name, seed, v, plans[v].drawn)


@pytest.mark.parametrize("angle", [0.0, 90.0, 180.0, 270.0, -90.0, 360.0, 45.0, -44.999, 1e-9, 89.99999, 133.7])
@pytest.mark.parametrize("hw", [(64, 64), (50, 72)])
def test_rotation_ops_equal_pil_rotate(angle, hw):
    """Image.rotate's fast paths (0 / 180 / quarter turns on squares) and the fixed-point affine path."""
    from PIL import Image
    from miseg_amd import slices as S
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, hw, dtype=np.uint8)
    want = np.array(Image.fromarray(img, mode="L").rotate(angle, False, False, None, fillcolor=0))
    h, w = hw
    plan = S.ViewPlan(S.rotation_ops(angle, w, h), [], w, h, {})
    got, _ = OA.run_jobs_numpy(S.encode_jobs([plan], [0]), img[None], img[None], h, w)
    assert np.array_equal((got[0] * 255).round().astype(np.uint8), want)


def test_blend_table_equals_pil_for_every_value_and_many_factors():
    from PIL import Image, ImageEnhance
    ramp = np.arange(256, dtype=np.uint8).reshape(16, 16)
    pil = Image.fromarray(ramp, mode="L")
    for f in np.concatenate([np.linspace(0.5, 1.5, 41), [0.0, 1.0, 0.999999, 1.000001, 2.5]]):
        alpha = np.float32(f)
        assert np.array_equal(OA._blend(0, ramp, alpha), np.array(ImageEnhance.Brightness(pil).enhance(float(f))))
        mean = int(float(ramp.astype(np.int64).sum()) / ramp.size + 0.5)
        assert np.array_equal(OA._blend(mean, ramp, alpha), np.array(ImageEnhance.Contrast(pil).enhance(float(f))))
        assert np.array_equal(ramp, np.array(ImageEnhance.Color(pil).enhance(float(f))))


def test_seed_plumbing_restatement():
    """plan_item draws exactly what the reference's wrappers draw (checked via the oracle's recorded colour ops)."""
    from miseg_amd import slices as S
    from semi_seg.augment import ACDCStrongTransforms as T
    tf = OA.acdc_transforms()["pretrain"]
    from PIL import Image
    img = Image.fromarray(np.zeros((256, 256), np.uint8), mode="L")
    for seed in (0, 5, 77777):
        tf(imgs=[img], targets=[img], global_seed=seed)
        last = tf._img_transform.transforms[0].last          # ops of view 2 (drawn last)
        plan = S.plan_item(T.pretrain, seed, 256, 256)[1]
        names = {S.BRIGHTNESS: "brightness", S.CONTRAST: "contrast", S.SATURATION: "saturation"}
        assert [(names[c], f) for c, f in plan.color] == last


def test_random_crop_too_small_raises():
    from miseg_amd import slices as S
    from semi_seg.augment import ACDCStrongTransforms as T
    with pytest.raises(ValueError):
        S.plan_item(T.pretrain, 1, 200, 256)


def test_acdc_index_split_partitions_and_val_selection(tmp_path):
    from semi_seg.synthetic import write_acdc_like
    from semi_seg import dataloader_helper as DH
    write_acdc_like(str(tmp_path), train_patients=10, val_patients=3, height=232, width=240)
    cfg = {"Data": {"name": "acdc", "labeled_data_ratio": 0.2, "unlabeled_data_ratio": 0.8},
           "LabeledData": {"shuffle": True, "batch_size": 3, "num_workers": 0},
           "UnlabeledData": {"shuffle": True, "batch_size": 5, "num_workers": 0}}
    lab, unlab, val = DH.get_dataloaders(cfg, root_dir=str(tmp_path), device="cpu")
    lg, ug = lab.dataset.show_group_set(), unlab.dataset.show_group_set()
    assert lg and ug and not (lg & ug) and len(lg) + len(ug) == 20          # patient-level split, ED/ES volumes are groups
    from sklearn.model_selection import train_test_split
    want_l, want_u = train_test_split(sorted(lg | ug), test_size=0.8, random_state=0)
    assert lg == set(want_l) and ug == set(want_u)
    assert len(val) == 6 and cfg["Data"]["name"] == "acdc"                   # config not mutated; 3 patients x 2 frames
    ds = unlab.dataset
    f = ds.get_filenames()[0]
    n = ds._acdc_info[ds._get_group_name(f)]
    parts = [ds._get_partition(f"{ds._get_group_name(f)}_{s:02d}") for s in range(n)]
    cut = n // 3
    assert parts == [str(0 if s <= cut - 1 else 1 if s <= 2 * cut else 2) for s in range(n)]
    v2 = DH.create_val_loader(unlab, val)
    assert len(v2) == 5 and v2.dataset.transform == val.dataset.transform
    assert v2.dataset.show_group_set() <= ug
    with pytest.raises(RuntimeError, match="no CPU path"):
        next(iter(lab))


def test_native_planner_equals_python_planner():
    """csrc/augment_plan.hip (CPython's MT19937 streams, Pillow's matrix set-up, in C++) against miseg_amd.slices.plan_item."""
    from miseg_amd import slices as S
    from semi_seg.augment import ACDCStrongTransforms as T
    rng = np.random.default_rng(0)
    for name in ("pretrain", "label", "val", "trainval"):
        rec = getattr(T, name)
        for (w, h) in [(256, 256), (224, 224), (301, 230), (224, 257)]:
            seeds = [0, 1, 2, 99999, 100000] + rng.integers(0, 100001, 60).tolist()
            ids = list(range(len(seeds)))
            jobs, ow, oh = S.plan_native(rec, seeds, ids, [w] * len(seeds), [h] * len(seeds))
            plans = [S.plan_item(rec, s, w, h) for s in seeds]
            views = 2 if rec.twice else 1
            want = S.encode_jobs([p[v] for v in range(views) for p in plans], ids * views)
            assert (ow, oh) == (224, 224) and jobs.shape == want.shape
            assert np.array_equal(jobs, want), (name, w, h, np.argwhere(jobs != want)[:5])
    big = S.Recipe(geo=(("rotate", 180),), twice=False)       # seeds beyond 32 bits use two key words, as random.seed does
    seeds = [2 ** 32 + 5, 2 ** 40 + 123, 2 ** 62]
    jobs, _, _ = S.plan_native(big, seeds, [0, 1, 2], [64] * 3, [64] * 3)
    assert np.array_equal(jobs, S.encode_jobs([S.plan_item(big, s, 64, 64)[0] for s in seeds], [0, 1, 2]))
    from miseg_amd._cabi import MisegError
    with pytest.raises(MisegError, match="RandomCrop"):
        S.plan_native(T.pretrain, [1], [0], [200], [256])
    with pytest.raises(MisegError, match="different output sizes"):
        S.plan_native(S.Recipe(geo=(("vflip", 0.5),)), [1, 2], [0, 1], [64, 65], [64, 64])
