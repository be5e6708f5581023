"""CPU oracle for the input pipeline.  TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Restates, on top of the real Pillow installed in the image, the transform chain the reference builds in
semi_seg/augment.py:7-52:

  * contrastyou/augment/sequential_wrapper.py:11-100   SequentialWrapper / SequentialWrapperTwice (seed plumbing)
  * whl:deepclustering2/decorator/decorator.py:196-212 FixRandomSeed
  * whl:deepclustering2/augment/pil_augment.py:111-360 RandomCrop, CenterCrop, RandomRotation, Random{Horizontal,Vertical}Flip,
                                                        ToTensor, ToLabel
  * torchvision==0.7.0 (requirement.txt:68; NOT under /root/reference and not installed here): ``transforms.Compose``,
    ``transforms.ColorJitter`` and the ``functional`` wrappers rotate / crop / center_crop / hflip / vflip / to_tensor /
    adjust_{brightness,contrast,saturation}, restated from its published source: thin wrappers over PIL calls
    (``img.rotate(angle, resample, expand, center, fillcolor=0)``, ``img.crop``, ``img.transpose``, ``ImageEnhance.*``);
    ColorJitter.get_params draws ``random.uniform`` per enabled factor in the order brightness, contrast, saturation, hue
    and then ``random.shuffle``s the list of adjustments.

Pinning: tests/golden/augment.npz was produced by running the REFERENCE's own SequentialWrapperTwice / pil_augment
classes (tests/golden/make_golden_augment.py) with this file's torchvision restatement plugged in where the reference
imports torchvision; test_oracle_golden checks this module against it.  The torchvision-0.7 pieces themselves are
therefore "parity unpinned" (restated, not executed); the Pillow arithmetic is the real library.

``run_jobs_numpy`` interprets the product's job table (include/miseg_hip.h, miseg_augment_slices) in numpy so that the
host planner / encoder can be checked against this oracle without a GPU.
"""
from __future__ import annotations

import random
from typing import List

import numpy as np
import torch
from PIL import Image, ImageEnhance


# ----------------------------------------------------------------------------- seeding (whl decorator.py:196-212)
class FixRandomSeed:
    """Seeds `random` and numpy's legacy generator for the block; afterwards both continue from their state at CONSTRUCTION time."""

    def __init__(self, random_seed: int = 0):
        self.random_seed = random_seed
        self._saved = {"py": random.getstate(), "np": np.random.get_state()}

    def __enter__(self):
        random.seed(self.random_seed)
        np.random.seed(self.random_seed)

    def __exit__(self, *_):
        random.setstate(self._saved["py"])
        np.random.set_state(self._saved["np"])


# ----------------------------------------------------------------------------- torchvision 0.7 functional (restated)
class tvf:
    @staticmethod
    def rotate(img, angle, resample=False, expand=False, center=None, fill=None):
        return img.rotate(angle, resample, expand, center, fillcolor=0 if fill is None else fill)

    @staticmethod
    def crop(img, top, left, height, width):
        return img.crop((left, top, left + width, top + height))

    @staticmethod
    def center_crop(img, output_size):
        if isinstance(output_size, (int, float)):
            output_size = (int(output_size), int(output_size))
        w, h = img.size
        th, tw = output_size
        return tvf.crop(img, int(round((h - th) / 2.0)), int(round((w - tw) / 2.0)), th, tw)

    @staticmethod
    def hflip(img):
        return img.transpose(Image.FLIP_LEFT_RIGHT)

    @staticmethod
    def vflip(img):
        return img.transpose(Image.FLIP_TOP_BOTTOM)

    @staticmethod
    def to_tensor(pic):
        assert pic.mode == "L", pic.mode
        img = torch.from_numpy(np.frombuffer(pic.tobytes(), dtype=np.uint8).copy()).view(pic.size[1], pic.size[0], 1)
        return img.permute(2, 0, 1).contiguous().float().div(255)

    @staticmethod
    def adjust_brightness(img, f):
        return ImageEnhance.Brightness(img).enhance(f)

    @staticmethod
    def adjust_contrast(img, f):
        return ImageEnhance.Contrast(img).enhance(f)

    @staticmethod
    def adjust_saturation(img, f):
        return ImageEnhance.Color(img).enhance(f)


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


class ColorJitter:
    """torchvision 0.7 transforms.ColorJitter for [lo, hi] range arguments, hue disabled."""

    def __init__(self, brightness=None, contrast=None, saturation=None):
        self.brightness, self.contrast, self.saturation = brightness, contrast, saturation
        self.last = None

    def __call__(self, img):
        ops = []
        if self.brightness is not None:
            f = random.uniform(self.brightness[0], self.brightness[1])
            ops.append(("brightness", f))
        if self.contrast is not None:
            f = random.uniform(self.contrast[0], self.contrast[1])
            ops.append(("contrast", f))
        if self.saturation is not None:
            f = random.uniform(self.saturation[0], self.saturation[1])
            ops.append(("saturation", f))
        random.shuffle(ops)
        self.last = list(ops)
        for name, f in ops:
            img = getattr(tvf, "adjust_" + name)(img, f)
        return img


class ToTensor:
    def __call__(self, pic):
        return pic if isinstance(pic, torch.Tensor) else tvf.to_tensor(pic)


# ----------------------------------------------------------------------------- pil_augment (whl pil_augment.py)
class RandomCrop:
    def __init__(self, size):
        self.size = (int(size), int(size))

    def __call__(self, img):
        w, h = img.size
        th, tw = self.size
        if w == tw and h == th:
            i, j = 0, 0
        else:
            i = random.randint(0, h - th)
            j = random.randint(0, w - tw)
        return tvf.crop(img, i, j, th, tw)


class CenterCrop:
    def __init__(self, size):
        self.size = (int(size), int(size))

    def __call__(self, img):
        return tvf.center_crop(img, self.size)


class RandomRotation:
    def __init__(self, degrees):
        self.degrees = (-degrees, degrees)

    def __call__(self, img):
        angle = random.uniform(self.degrees[0], self.degrees[1])
        return tvf.rotate(img, angle, False, False, None)


class RandomHorizontalFlip:
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, img):
        return tvf.hflip(img) if random.random() < self.p else img


class RandomVerticalFlip:
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, img):
        return tvf.vflip(img) if random.random() < self.p else img


class ToLabel:
    def __call__(self, img):
        return torch.from_numpy(np.array(img)[None, ...].astype(np.float32)).long()


# ----------------------------------------------------------------------------- sequential_wrapper.py:11-100
class SequentialWrapper:
    def __init__(self, comm_transform=None, img_transform=None, target_transform=None):
        self._comm_transform = comm_transform
        self._img_transform = img_transform if img_transform is not None else ToTensor()
        self._target_transform = target_transform if target_transform is not None else ToLabel()

    def __call__(self, imgs, targets=None, comm_seed=None, img_seed=None, target_seed=None):
        _comm_seed = int(random.randint(0, int(1e5))) if comm_seed is None else int(comm_seed)
        imgs_c, targets_c = imgs, targets
        if self._comm_transform:
            imgs_c, targets_c = [], []
            for img in imgs:
                with FixRandomSeed(_comm_seed):
                    imgs_c.append(self._comm_transform(img))
            if targets:
                for t in targets:
                    with FixRandomSeed(_comm_seed):
                        targets_c.append(self._comm_transform(t))
        out_i, out_t = [], []
        _img_seed = int(random.randint(0, int(1e5))) if img_seed is None else int(img_seed)
        for img in imgs_c:
            with FixRandomSeed(_img_seed):
                out_i.append(self._img_transform(img))
        _target_seed = int(random.randint(0, int(1e5))) if target_seed is None else int(target_seed)
        if targets_c:
            for t in targets_c:
                with FixRandomSeed(_target_seed):
                    out_t.append(self._target_transform(t))
        if targets is None:
            return out_i
        return [*out_i, *out_t]


class SequentialWrapperTwice(SequentialWrapper):
    def __init__(self, comm_transform=None, img_transform=None, target_transform=None, total_freedom=True):
        super().__init__(comm_transform, img_transform, target_transform)
        self._total_freedom = total_freedom

    def __call__(self, imgs, targets=None, global_seed=None, **kwargs):
        global_seed = int(random.randint(0, int(1e5))) if global_seed is None else int(global_seed)
        with FixRandomSeed(global_seed):
            c1, c2 = int(random.randint(0, int(1e5))), int(random.randint(0, int(1e5)))
            i1, i2 = int(random.randint(0, int(1e5))), int(random.randint(0, int(1e5)))
            t1, t2 = int(random.randint(0, int(1e5))), int(random.randint(0, int(1e5)))
            if self._total_freedom:
                return [super().__call__(imgs, targets, c1, i1, t1), super().__call__(imgs, targets, c2, i2, t2)]
            return [super().__call__(imgs, targets, c1, i1, t1), super().__call__(imgs, targets, c1, i2, t1)]


# ----------------------------------------------------------------------------- semi_seg/augment.py:7-52
def _jitter():
    return Compose([ColorJitter(brightness=[0.5, 1.5], contrast=[0.5, 1.5], saturation=[0.5, 1.5]), ToTensor()])


def acdc_transforms():
    return {
        "pretrain": SequentialWrapperTwice(
            comm_transform=Compose([RandomRotation(45), RandomVerticalFlip(), RandomHorizontalFlip(), RandomCrop(224)]),
            img_transform=_jitter(), target_transform=Compose([ToLabel()]), total_freedom=True),
        "label": SequentialWrapperTwice(comm_transform=Compose([RandomCrop(224), RandomRotation(30)]),
                                        img_transform=Compose([ToTensor()]), target_transform=Compose([ToLabel()])),
        "val": SequentialWrapper(comm_transform=CenterCrop(224)),
        "trainval": SequentialWrapperTwice(comm_transform=Compose([RandomCrop(224)]), img_transform=Compose([ToTensor()]),
                                           target_transform=Compose([ToLabel()]), total_freedom=True),
    }


def apply(name: str, img: np.ndarray, gt: np.ndarray, global_seed: int):
    """One dataset item through preset ``name``: u8 arrays in, what ACDCDataset.__getitem__ returns as ``data`` out."""
    tf = acdc_transforms()[name]
    pi, pg = Image.fromarray(img, mode="L"), Image.fromarray(gt, mode="L")
    if isinstance(tf, SequentialWrapperTwice):
        return tf(imgs=[pi], targets=[pg], global_seed=global_seed)
    return tf(imgs=[pi], targets=[pg])


# ----------------------------------------------------------------------------- numpy interpreter of the job table
def _blend(base: int, v: np.ndarray, alpha: np.float32) -> np.ndarray:
    """libImaging Blend.c on one band."""
    t = np.float32(base) + alpha * (v.astype(np.int32) - np.int32(base)).astype(np.float32)
    t = t.astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32))).astype(np.uint8)


def run_jobs_numpy(jobs: np.ndarray, atlas_img: np.ndarray, atlas_gt: np.ndarray, out_h: int, out_w: int):
    """Same contract as miseg_augment_slices (include/miseg_hip.h), in numpy: returns (img fp32 [n,H,W], gt int64 [n,H,W])."""
    n = jobs.shape[0]
    img_out = np.zeros((n, out_h, out_w), np.float32)
    gt_out = np.zeros((n, out_h, out_w), np.int64)
    for r in range(n):
        job = jobs[r].astype(np.int64)
        y, x = np.meshgrid(np.arange(out_h, dtype=np.int64), np.arange(out_w, dtype=np.int64), indexing="ij")
        ok = np.ones((out_h, out_w), bool)
        for g in range(int(job[3]) - 1, -1, -1):
            op = job[12 + 9 * g: 12 + 9 * g + 9]
            iw, ih = op[7], op[8]
            if op[0] == 1:
                y, x = y + op[1], x + op[2]
                ok &= (y >= 0) & (y < ih) & (x >= 0) & (x < iw)
            elif op[0] == 2:
                y = ih - 1 - y
            elif op[0] == 3:
                x = iw - 1 - x
            elif op[0] == 4:
                xx = (op[3] + y * op[2] + x * op[1]).astype(np.int32)
                yy = (op[6] + y * op[5] + x * op[4]).astype(np.int32)
                x, y = (xx >> 16).astype(np.int64), (yy >> 16).astype(np.int64)
                ok &= (y >= 0) & (y < ih) & (x >= 0) & (x < iw)
            y, x = np.where(ok, y, 0), np.where(ok, x, 0)
        v = np.where(ok, atlas_img[job[0], y, x], 0).astype(np.uint8)
        gt_out[r] = np.where(ok, atlas_gt[job[0], y, x], 0)
        for c in range(int(job[4])):
            code = int(job[5 + c])
            alpha = np.array([jobs[r, 8 + c]], dtype=np.int32).view(np.float32)[0]
            if code == 1:
                v = _blend(0, v, alpha)
            elif code == 2:
                mean = float(v.astype(np.int64).sum()) / float(v.size)
                v = _blend(int(mean + 0.5), v, alpha)
        img_out[r] = v.astype(np.float32) / np.float32(255.0)
    return img_out, gt_out
