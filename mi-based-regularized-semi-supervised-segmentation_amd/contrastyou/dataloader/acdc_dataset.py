"""ACDC slice index + semi-supervised split with the reference's names and semantics, backed by the device-resident
pipeline (``miseg_amd.slices``) instead of per-item PIL work in DataLoader workers.

ref: contrastyou/dataloader/acdc_dataset.py:14-60 (ACDCDataset: img/gt sub-folders, ``acdc_info.npy``, group =
``patientNNN_FF``, partition = third of the volume the slice index falls in), whl:deepclustering2/dataset/segmentation/
_medicalSegmentationDataset.py:30-209 (file discovery: sorted png/jpg per sub-folder, equal counts), acdc_dataset.py:56-134
(patient-level ``train_test_split(groups, test_size=unlabeled_ratio, random_state=seed)``), _patient_sampler.py:86-104
(sub-dataset by patient list).  On-disk layout: ``<root>/ACDC_contrast/{train,val}/{img,gt}/patientNNN_FF_SS.png``.
"""
from __future__ import annotations

import os
import re
from copy import copy
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from miseg_amd import slices as S


class ACDCDataset:
    folder_name = "ACDC_contrast"
    dataset_modes = ["train", "val", "test", "unlabeled"]
    allow_extension = [".jpg", ".png"]

    def __init__(self, root_dir: str, mode: str, transforms: Optional[S.Recipe] = None, verbose: bool = True, device=None,
                 *args, **kwargs) -> None:
        assert mode in self.dataset_modes, mode
        self._root_dir = os.path.join(root_dir, self.folder_name)
        if not Path(self._root_dir).is_dir():
            raise FileNotFoundError(f"{self._root_dir}: ACDC_contrast not found (no network here: place the unzipped dataset there)")
        self._mode, self._name, self._subfolders = mode, f"{mode}_dataset", ["img", "gt"]
        self._verbose = verbose
        self._filenames = self._make_dataset(self._root_dir, mode, self._subfolders, verbose)
        self._pattern = r"patient\d+_\d+"
        self._re_pattern = re.compile(self._pattern)
        self._acdc_info = np.load(os.path.join(self._root_dir, "acdc_info.npy"), allow_pickle=True).item()
        assert isinstance(self._acdc_info, dict) and len(self._acdc_info) == 200
        self._transform = transforms if transforms is not None else S.Recipe()
        self._device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self._resident: Optional[S.ResidentSlices] = None

    # ---- file index (whl _medicalSegmentationDataset.py:174-209)
    @classmethod
    def _make_dataset(cls, root: str, mode: str, subfolders: List[str], verbose=True) -> Dict[str, List[str]]:
        imgs = {}
        for sub in subfolders:
            d = Path(root, mode, sub)
            assert d.is_dir(), str(d)
            names = [x for x in os.listdir(d) if Path(x).suffixes[:1] and Path(x).suffixes[0] in cls.allow_extension]
            imgs[sub] = sorted(os.path.join(root, mode, sub, x) for x in names)
        assert len({len(v) for v in imgs.values()}) == 1, {k: len(v) for k, v in imgs.items()}
        if verbose:
            for sub in subfolders:
                print(f"found {len(imgs[sub])} images in {sub}\t")
        return imgs

    @property
    def dataset_pattern(self) -> str:
        return self._pattern

    @property
    def mode(self) -> str:
        return self._mode

    @property
    def transform(self):
        return self._transform

    def set_transform(self, transform: S.Recipe) -> None:
        self._transform = transform

    def get_filenames(self, subfolder_name=None) -> List[str]:
        return self._filenames[subfolder_name or self._subfolders[0]]

    def __len__(self) -> int:
        return len(self._filenames[self._subfolders[0]])

    # ---- grouping (contrastyou acdc_dataset.py:37-56)
    def _get_group_name(self, path) -> str:
        m = self._re_pattern.search(Path(path).stem)
        if m is None:
            raise AttributeError(f"Cannot match pattern: {self._pattern} for path: {path}")
        return m.group(0)

    def get_group_list(self) -> List[str]:
        return sorted({self._get_group_name(f) for f in self.get_filenames()})

    def _get_group(self, filename) -> str:
        return str(self._get_group_name(filename))

    def _get_partition(self, filename) -> str:
        cutting_point = self._acdc_info[self._get_group_name(filename)] // 3
        cur_index = int(re.compile(r"\d+").findall(Path(filename).stem)[-1])
        if cur_index <= cutting_point - 1:
            return str(0)
        if cur_index <= 2 * cutting_point:
            return str(1)
        return str(2)

    def show_paritions(self) -> List[str]:
        return [self._get_partition(f) for f in self.get_filenames()]

    def show_groups(self) -> List[str]:
        return [self._get_group(f) for f in self.get_filenames()]

    def show_parition_set(self):
        return set(self.show_paritions())

    def show_group_set(self):
        return set(self.show_groups())

    # ---- device side
    def resident(self) -> S.ResidentSlices:
        if self._resident is None:
            self._resident = S.ResidentSlices(self._filenames["img"], self._filenames["gt"], self._device)
        return self._resident

    def subset(self, keep: Sequence[bool]) -> "ACDCDataset":
        """Same dataset restricted to the flagged files (deep-copy semantics of the reference's sub-dataset helpers);
        the decoded atlases are dropped and rebuilt lazily for the subset."""
        other = copy(self)
        other._filenames = {k: [f for f, ok in zip(v, keep) if ok] for k, v in self._filenames.items()}
        other._resident = None
        return other

    def collate(self, indices: Sequence[int], item_seeds: Sequence[int]):
        """What default_collate makes of ``[self[i] for i in indices]`` in the reference (acdc_dataset.py:26-33): a list per
        view of [img batch, target batch], then the filename / partition / group lists -- produced by one launch."""
        res = self.resident()
        sizes = [res.sizes[i] for i in indices]
        jobs, ow, oh = S.plan_native(self._transform, item_seeds, indices, [s[0] for s in sizes], [s[1] for s in sizes])
        img, gt = res.run(jobs, ow, oh)
        n = len(indices)
        data = [[img[v * n:(v + 1) * n], gt[v * n:(v + 1) * n]] for v in range(2 if self._transform.twice else 1)]
        if not self._transform.twice:
            data = data[0]
        names, parts, groups = self._meta()
        return data, [names[i] for i in indices], [parts[i] for i in indices], [groups[i] for i in indices]

    def _meta(self):
        """(stem, partition, group) per file, computed once per file list."""
        key = id(self._filenames)
        if getattr(self, "_meta_key", None) != key:
            names = [Path(f).stem for f in self._filenames["img"]]
            self._meta_cache = (names, [self._get_partition(f) for f in names], [self._get_group(f) for f in names])
            self._meta_key = key
        return self._meta_cache

    def __getitem__(self, index) -> Tuple[list, str, str, str]:
        data, names, parts, groups = self.collate([index], [int(np.random.randint(0, int(1e5)))])
        squeeze = (lambda t: t[0])
        data = [[squeeze(t) for t in view] for view in data] if self._transform.twice else [squeeze(t) for t in data]
        return data, names[0], parts[0], groups[0]


class ACDCSemiInterface:
    """ref contrastyou/dataloader/acdc_dataset.py:55-60 + whl acdc_dataset.py:56-134, semi_helper.py:344-370."""

    def __init__(self, root_dir, labeled_data_ratio: float = 0.2, unlabeled_data_ratio: float = 0.8, seed: int = 0,
                 verbose: bool = True, device=None) -> None:
        assert (labeled_data_ratio + unlabeled_data_ratio) <= 1, \
            f"`labeled_data_ratio` + `unlabeled_data_ratio` should be less than 1.0, given {labeled_data_ratio + unlabeled_data_ratio}"
        self.DataClass = ACDCDataset
        self.root_dir, self.seed, self.verbose, self.device = root_dir, seed, verbose, device
        self.labeled_ratio, self.unlabeled_ratio = labeled_data_ratio, unlabeled_data_ratio
        self.val_ratio = 1 - (labeled_data_ratio + unlabeled_data_ratio)

    def _create_semi_supervised_datasets(self, labeled_transform=None, unlabeled_transform=None, val_transform=None):
        from sklearn.model_selection import train_test_split
        train_set = self.DataClass(root_dir=self.root_dir, mode="train", verbose=self.verbose, device=self.device)
        val_set = self.DataClass(root_dir=self.root_dir, mode="val", verbose=self.verbose, device=self.device)
        if self.labeled_ratio == 1 or self.unlabeled_ratio == 1:
            labeled_set, unlabeled_set = train_set, train_set.subset([True] * len(train_set))
        else:
            labeled_patients, unlabeled_patients = train_test_split(train_set.get_group_list(), test_size=self.unlabeled_ratio,
                                                                    random_state=self.seed)
            files = train_set.get_filenames()
            labeled_set = train_set.subset([train_set._get_group_name(f) in labeled_patients for f in files])
            unlabeled_set = train_set.subset([train_set._get_group_name(f) in unlabeled_patients for f in files])
            assert len(labeled_set) + len(unlabeled_set) == len(train_set), "wrong on labeled/unlabeled split."
        for ds, tf in ((labeled_set, labeled_transform), (unlabeled_set, unlabeled_transform), (val_set, val_transform)):
            if tf:
                ds.set_transform(tf)
        return labeled_set, unlabeled_set, val_set
