"""Device-resident input pipeline.

The reference feeds the step from DataLoader worker processes that open a PNG pair per slice and push it through a PIL
transform chain (semi_seg/dataloader_helper.py:23-73, semi_seg/augment.py:7-52, contrastyou/augment/sequential_wrapper.py,
whl:deepclustering2/augment/pil_augment.py).  ACDC is ~2k slices of <= 256^2 bytes: on a 288 GB part the whole dataset is a
rounding error, so here it is decoded ONCE into two u8 atlases in HBM and every batch is produced by one kernel launch
(csrc/augment.hip) from a small table of augmentation jobs.  The host only draws the random parameters -- from the same
``random`` streams, in the same order, as the reference's transform objects would consume them -- so for a given item seed
the batch is bit-identical to what the PIL chain returns.

  Recipe            declarative form of a SequentialWrapper / SequentialWrapperTwice
  plan_item()       item seed + slice size -> per-view geometric / colour op lists (the random-stream restatement)
  encode_jobs()     op lists -> the int32 job table of include/miseg_hip.h (miseg_augment_slices)
  ResidentSlices    the atlases
  AugmentedLoader   infinite sampler + planner + launch; yields the reference's collated batch structure
  PatientLoader     one batch per patient (validation / test), reference PatientSampler order
"""
from __future__ import annotations

import math
import random
import re
import struct
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _cabi

JOB_INTS, MAX_GEO = 48, 4
CROP, VFLIP, HFLIP, AFFINE = 1, 2, 3, 4
BRIGHTNESS, CONTRAST, SATURATION = 1, 2, 3


# ------------------------------------------------------------------------------------------ recipes
@dataclass(frozen=True)
class Recipe:
    """``geo``: tuple of ("rotate", degrees) | ("vflip", p) | ("hflip", p) | ("random_crop", size) | ("center_crop", size)
    in application order (the comm_transform); ``jitter``: (brightness, contrast, saturation) ranges of a torchvision-0.7
    ColorJitter or None (the img_transform, always followed by ToTensor; targets get ToLabel); ``twice`` /
    ``total_freedom``: SequentialWrapperTwice semantics (sequential_wrapper.py:72-100)."""
    geo: Tuple[Tuple[str, float], ...] = ()
    jitter: Optional[Tuple[Tuple[float, float], Tuple[float, float], Tuple[float, float]]] = None
    twice: bool = False
    total_freedom: bool = True

    def out_size(self, w: int, h: int) -> Tuple[int, int]:
        for kind, arg in self.geo:
            if kind in ("random_crop", "center_crop"):
                w = h = int(arg)
        return w, h


@dataclass
class ViewPlan:
    geo: List[Tuple[int, ...]]        # (type, p1..p6, in_w, in_h)
    color: List[Tuple[int, float]]    # (code, factor) in application order
    out_w: int
    out_h: int
    drawn: Dict[str, object]          # the raw draws (angle, flips, crop origin, factors) for tests / logging


def _fix(v: float) -> int:
    """libImaging Geometry.c FIX(): FLOOR(v * 65536 + 0.5) with FLOOR = C truncation for v >= 0, floor() below."""
    v = v * 65536.0 + 0.5
    return int(math.floor(v)) if v < 0.0 else int(v)


def _wrap32(v: int) -> int:
    return ((v + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)


def rotation_ops(angle: float, w: int, h: int) -> List[Tuple[int, ...]]:
    """PIL ``Image.rotate(angle, NEAREST, expand=False, center=None, fillcolor=0)`` as job ops.
    Pillow Image.py (rotate): the angle is reduced mod 360; 0 / 180 / (90, 270 on squares) take the transpose fast
    paths; otherwise matrix = [cos, sin, 0, -sin, cos, 0] of -angle, rounded to 15 digits, about the centre (w/2, h/2),
    handed to libImaging's affine_fixed (16.16 fixed point with the half-pixel folded into a2 / a5)."""
    angle = angle % 360.0
    one = 65536
    if angle == 0:
        return []
    if angle == 180:
        return [(VFLIP, 0, 0, 0, 0, 0, 0, w, h), (HFLIP, 0, 0, 0, 0, 0, 0, w, h)]
    if angle == 90 and w == h:      # counter-clockwise quarter turn: in = (w-1-y, x)
        return [(AFFINE, 0, -one, (w - 1) * one + one // 2, one, 0, one // 2, w, h)]
    if angle == 270 and w == h:     # in = (y, h-1-x)
        return [(AFFINE, 0, one, one // 2, -one, 0, (h - 1) * one + one // 2, w, h)]
    rad = -math.radians(angle)
    m = [round(math.cos(rad), 15), round(math.sin(rad), 15), 0.0, round(-math.sin(rad), 15), round(math.cos(rad), 15), 0.0]
    cx, cy = w / 2.0, h / 2.0
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5]
    m[2] += cx
    m[5] += cy
    a0, a1, a3, a4 = _fix(m[0]), _fix(m[1]), _fix(m[3]), _fix(m[4])
    a2 = _fix(m[2] + m[0] * 0.5 + m[1] * 0.5)
    a5 = _fix(m[5] + m[3] * 0.5 + m[4] * 0.5)
    return [(AFFINE, a0, a1, a2, a3, a4, a5, w, h)]


def plan_view(recipe: Recipe, comm_seed: int, img_seed: int, w: int, h: int) -> ViewPlan:
    """One SequentialWrapper.__call__ (sequential_wrapper.py:27-62): the comm transform runs under FixRandomSeed(comm_seed)
    (same draws for image and target), the image transform under FixRandomSeed(img_seed)."""
    rng = random.Random(comm_seed)
    geo: List[Tuple[int, ...]] = []
    drawn: Dict[str, object] = {}
    for kind, arg in recipe.geo:
        if kind == "rotate":          # pil_augment.RandomRotation.get_params: random.uniform(-d, d)
            angle = rng.uniform(-arg, arg)
            drawn["angle"] = angle
            geo += rotation_ops(angle, w, h)
        elif kind == "vflip":         # random.random() < p
            flip = rng.random() < arg
            drawn["vflip"] = flip
            if flip:
                geo.append((VFLIP, 0, 0, 0, 0, 0, 0, w, h))
        elif kind == "hflip":
            flip = rng.random() < arg
            drawn["hflip"] = flip
            if flip:
                geo.append((HFLIP, 0, 0, 0, 0, 0, 0, w, h))
        elif kind == "random_crop":   # pil_augment.RandomCrop.get_params: no draw when the size already matches
            th = tw = int(arg)
            if w == tw and h == th:
                i = j = 0
            else:
                if h < th or w < tw:
                    raise ValueError(f"RandomCrop({th}) on a {w}x{h} slice")  # random.randint raises in the reference too
                i = rng.randint(0, h - th)
                j = rng.randint(0, w - tw)
            drawn["crop"] = (i, j)
            geo.append((CROP, i, j, 0, 0, 0, 0, w, h))
            w, h = tw, th
        elif kind == "center_crop":   # torchvision 0.7 F.center_crop: int(round((h - th) / 2.))
            th = tw = int(arg)
            i, j = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
            drawn["crop"] = (i, j)
            geo.append((CROP, i, j, 0, 0, 0, 0, w, h))
            w, h = tw, th
        else:
            raise ValueError(kind)
    if len(geo) > MAX_GEO:
        raise ValueError(f"{len(geo)} geometric ops, the job table holds {MAX_GEO}")
    color: List[Tuple[int, float]] = []
    if recipe.jitter is not None:     # torchvision 0.7 ColorJitter.get_params: three uniforms, then random.shuffle
        rng = random.Random(img_seed)
        ops = [(code, rng.uniform(lo, hi)) for code, (lo, hi) in zip((BRIGHTNESS, CONTRAST, SATURATION), recipe.jitter)]
        rng.shuffle(ops)
        color = ops
        drawn["color"] = list(ops)
    return ViewPlan(geo, color, w, h, drawn)


def plan_item(recipe: Recipe, item_seed: int, w: int, h: int) -> List[ViewPlan]:
    """SequentialWrapperTwice.__call__ (sequential_wrapper.py:86-100): six seeds from FixRandomSeed(global_seed), two views;
    a plain SequentialWrapper draws nothing that matters for a seed-free chain (CenterCrop)."""
    if not recipe.twice:
        rng = random.Random(item_seed)
        comm, img = int(rng.randint(0, int(1e5))), int(rng.randint(0, int(1e5)))
        return [plan_view(recipe, comm, img, w, h)]
    rng = random.Random(item_seed)
    comm1, comm2 = int(rng.randint(0, int(1e5))), int(rng.randint(0, int(1e5)))
    img1, img2 = int(rng.randint(0, int(1e5))), int(rng.randint(0, int(1e5)))
    _t1, _t2 = int(rng.randint(0, int(1e5))), int(rng.randint(0, int(1e5)))
    if recipe.total_freedom:
        return [plan_view(recipe, comm1, img1, w, h), plan_view(recipe, comm2, img2, w, h)]
    return [plan_view(recipe, comm1, img1, w, h), plan_view(recipe, comm1, img2, w, h)]


def encode_jobs(plans: Sequence[ViewPlan], slices: Sequence[int]) -> np.ndarray:
    jobs = np.zeros((len(plans), JOB_INTS), dtype=np.int32)
    for r, (plan, sl) in enumerate(zip(plans, slices)):
        row = jobs[r]
        row[0], row[3], row[4] = sl, len(plan.geo), len(plan.color)
        for c, (code, factor) in enumerate(plan.color):
            row[5 + c] = code
            row[8 + c] = struct.unpack("<i", struct.pack("<f", factor))[0]   # the C side takes alpha as a float
        for g, op in enumerate(plan.geo):
            row[12 + 9 * g: 12 + 9 * g + 9] = [_wrap32(int(v)) for v in op]
    return jobs


# ------------------------------------------------------------------------------------------ native planner
import ctypes  # noqa: E402

_KIND = {"rotate": 1, "vflip": 2, "hflip": 3, "random_crop": 4, "center_crop": 5}


class _CRecipe(ctypes.Structure):   # miseg_aug_recipe (include/miseg_hip.h)
    _fields_ = [("n_geo", ctypes.c_int32), ("geo_kind", ctypes.c_int32 * 4), ("geo_arg", ctypes.c_double * 4),
                ("has_jitter", ctypes.c_int32), ("jitter", ctypes.c_double * 6), ("twice", ctypes.c_int32),
                ("total_freedom", ctypes.c_int32)]


def _c_recipe(recipe: Recipe) -> _CRecipe:
    if len(recipe.geo) > 4:
        raise ValueError("at most 4 geometric transforms")
    c = _CRecipe()
    c.n_geo = len(recipe.geo)
    for g, (kind, arg) in enumerate(recipe.geo):
        c.geo_kind[g], c.geo_arg[g] = _KIND[kind], float(arg)
    c.has_jitter = int(recipe.jitter is not None)
    if recipe.jitter is not None:
        for k, (lo, hi) in enumerate(recipe.jitter):
            c.jitter[2 * k], c.jitter[2 * k + 1] = float(lo), float(hi)
    c.twice, c.total_freedom = int(recipe.twice), int(recipe.total_freedom)
    return c


_RECIPES: Dict[Recipe, _CRecipe] = {}


def plan_native(recipe: Recipe, item_seeds: Sequence[int], slice_ids: Sequence[int], widths: Sequence[int],
                heights: Sequence[int]) -> Tuple[np.ndarray, int, int]:
    """The whole batch planned by the C++ twin of plan_item + encode_jobs (csrc/augment_plan.hip, miseg_plan_augment):
    returns (jobs int32 [views * n, JOB_INTS] view-major, out_w, out_h)."""
    c = _RECIPES.get(recipe)
    if c is None:
        c = _RECIPES[recipe] = _c_recipe(recipe)
    n = len(item_seeds)
    views = 2 if recipe.twice else 1
    seeds = np.asarray(item_seeds, dtype=np.int64)
    ids, ws, hs = (np.asarray(a, dtype=np.int32) for a in (slice_ids, widths, heights))
    jobs = np.empty((views * n, JOB_INTS), dtype=np.int32)
    wh = np.zeros(2, dtype=np.int32)
    _cabi.call("miseg_plan_augment", ctypes.addressof(c), n, seeds.ctypes.data, ids.ctypes.data, ws.ctypes.data, hs.ctypes.data,
               jobs.ctypes.data, wh.ctypes.data)
    return jobs, int(wh[0]), int(wh[1])


# ------------------------------------------------------------------------------------------ the resident dataset
class ResidentSlices:
    """Every slice of a dataset decoded once into u8 atlases [N, Hmax, Wmax] on the device (image and ground truth)."""

    def __init__(self, img_paths: Sequence[str], gt_paths: Optional[Sequence[str]], device):
        from PIL import Image
        if len(img_paths) == 0:
            raise ValueError("empty dataset")
        imgs, gts, self.sizes = [], [], []
        for k, p in enumerate(img_paths):
            with Image.open(p) as im:
                if im.mode != "L":
                    raise ValueError(f"{p}: mode {im.mode}; the device pipeline restates the PIL chain for 8-bit 'L' slices only")
                a = np.array(im, dtype=np.uint8)
            imgs.append(a)
            self.sizes.append((a.shape[1], a.shape[0]))
            if gt_paths is not None:
                with Image.open(gt_paths[k]) as im:
                    if im.mode != "L":
                        raise ValueError(f"{gt_paths[k]}: mode {im.mode}, expected 'L'")
                    g = np.array(im, dtype=np.uint8)
                if g.shape != a.shape:
                    raise ValueError(f"{gt_paths[k]}: {g.shape} vs image {a.shape}")
                gts.append(g)
        self.h = max(a.shape[0] for a in imgs)
        self.w = max(a.shape[1] for a in imgs)
        self.device = torch.device(device)
        self.img = self._atlas(imgs)
        self.gt = self._atlas(gts) if gt_paths is not None else None

    @classmethod
    def from_arrays(cls, imgs: Sequence[np.ndarray], gts: Optional[Sequence[np.ndarray]], device) -> "ResidentSlices":
        self = cls.__new__(cls)
        self.sizes = [(a.shape[1], a.shape[0]) for a in imgs]
        self.h, self.w = max(a.shape[0] for a in imgs), max(a.shape[1] for a in imgs)
        self.device = torch.device(device)
        self.img = self._atlas(imgs)
        self.gt = self._atlas(gts) if gts is not None else None
        return self

    def _atlas(self, arrays) -> torch.Tensor:
        atlas = np.zeros((len(arrays), self.h, self.w), dtype=np.uint8)
        for k, a in enumerate(arrays):
            atlas[k, :a.shape[0], :a.shape[1]] = a
        return torch.from_numpy(atlas).to(self.device)

    def __len__(self) -> int:
        return len(self.sizes)

    def run(self, jobs: np.ndarray, out_w: int, out_h: int) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """jobs int32 [n, JOB_INTS] (host) -> img fp32 [n,1,H,W], labels int64 [n,1,H,W] on the device."""
        if not self.img.is_cuda:
            raise RuntimeError("the input pipeline runs on the GPU (csrc/augment.hip); there is no CPU path")
        n = int(jobs.shape[0])
        # the job table goes up from PINNED memory, non-blocking (eight slots round-robin: two loaders x up to two iterations of host
        # run-ahead x safety): a pageable `.to(device)` is a synchronous copy, i.e. the host would wait for the GPU queue to drain
        # once per batch (semi_seg/epocher.py `_upload_flips` has the measurement)
        jh = torch.from_numpy(np.ascontiguousarray(jobs, dtype=np.int32))
        from .ops import PinnedRing
        rings = self.__dict__.setdefault("_job_ring", {})
        key = tuple(jh.shape)
        ring = rings.get(key)
        if ring is None:
            ring = rings[key] = PinnedRing(key, torch.int32, slots=8)
        jd = ring.upload(lambda slot: slot.copy_(jh), self.device)
        img = torch.empty(n, 1, out_h, out_w, dtype=torch.float32, device=self.device)
        gt = torch.empty(n, 1, out_h, out_w, dtype=torch.int64, device=self.device) if self.gt is not None else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _cabi.call("miseg_augment_slices", stream, self.img.data_ptr(), self.gt.data_ptr() if self.gt is not None else None,
                   len(self), self.h, self.w, jd.data_ptr(), n, out_h, out_w, img.data_ptr(), gt.data_ptr() if gt is not None else None)
        return img, gt


# ------------------------------------------------------------------------------------------ loaders
class AugmentedLoader:
    """Infinite loader: reference = DataLoader(dataset, sampler=InfiniteRandomSampler(shuffle), batch_size) with the
    dataset's SequentialWrapperTwice (dataloader_helper.py:40-58; whl:deepclustering2/dataloader/sampler.py:199-236).
    Yields ``[[img, tgt], [img2, tgt2]], filenames, partitions, groups`` with device tensors."""

    def __init__(self, dataset, batch_size: int, shuffle: bool = True, seed: int = 0):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle = bool(shuffle)
        self._gen = torch.Generator().manual_seed(int(seed))
        self._rng = random.Random(int(seed) * 7919 + 1)
        self._order: List[int] = []

    def __iter__(self):
        return self

    def __len__(self):
        return len(self.dataset)

    def _indices(self) -> List[int]:
        out = []
        while len(out) < self.batch_size:
            if not self._order:
                n = len(self.dataset)
                self._order = (torch.randperm(n, generator=self._gen) if self.shuffle else torch.arange(n)).tolist()
            out.append(self._order.pop(0))
        return out

    def __next__(self):
        idx = self._indices()
        seeds = [int(self._rng.randint(0, int(1e5))) for _ in idx]   # SequentialWrapperTwice's global_seed draw
        return self.dataset.collate(idx, seeds)


class PatientLoader:
    """Finite loader, one batch = every slice of one patient in file order, patients sorted
    (reference: DataLoader(batch_sampler=PatientSampler(dataset, grp_regex, shuffle=False)), dataloader_helper.py:60-72)."""

    def __init__(self, dataset):
        self.dataset = dataset
        groups: Dict[str, List[int]] = {}
        for i, f in enumerate(dataset.get_filenames()):
            groups.setdefault(dataset._get_group_name(f), []).append(i)
        self._batches = [groups[k] for k in sorted(groups)]

    def __len__(self):
        return len(self._batches)

    def __iter__(self):
        for idx in self._batches:
            yield self.dataset.collate(idx, [0] * len(idx))


_PATIENT = re.compile(r"patient\d+_\d+")


def stem(path: str) -> str:
    return Path(path).stem
