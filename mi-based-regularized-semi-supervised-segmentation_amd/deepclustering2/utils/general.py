"""Assert / one-hot / seeding helpers (ref whl:deepclustering2/utils/general.py:64-81,145-196,221-235,285-330)."""
from __future__ import annotations

import collections.abc
import os
import random
import subprocess
from typing import Any, Dict, Iterable, Set

import numpy as np
import torch
from torch import Tensor

__all__ = ["uniq", "sset", "simplex", "one_hot", "class2one_hot", "probs2class", "probs2one_hot", "set_benchmark", "fix_all_seed",
           "iter_average", "dict_merge", "flatten_dict", "nice_dict", "gethash", "assert_list", "to_float", "path2Path", "write_yaml"]


def uniq(a: Tensor) -> Set:
    return set(v.item() for v in a.unique())


def sset(a: Tensor, sub: Iterable) -> bool:
    return uniq(a).issubset(sub)


def simplex(t: Tensor, axis=1) -> bool:
    s = t.sum(axis).type(torch.float32)
    return bool(torch.allclose(s, torch.ones_like(s), rtol=1e-4, atol=1e-4))


def one_hot(t: Tensor, axis=1) -> bool:
    return simplex(t, axis) and sset(t, [0, 1])


def class2one_hot(seg: Tensor, C: int, class_dim: int = 1) -> Tensor:
    """int64 one-hot along ``class_dim``; labels outside [0,C) raise AssertionError like the reference."""
    if seg.dim() == 2:
        seg = seg.unsqueeze(0)
    assert sset(seg, list(range(C)))
    res = torch.stack([seg == c for c in range(C)], dim=class_dim).type(torch.long)
    return res


def probs2class(probs: Tensor, class_dim: int = 1) -> Tensor:
    assert simplex(probs, axis=class_dim)
    return probs.argmax(dim=class_dim)


def probs2one_hot(probs: Tensor, class_dim: int = 1) -> Tensor:
    return class2one_hot(probs2class(probs, class_dim), probs.shape[class_dim], class_dim)


def fix_all_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def set_benchmark(seed):
    """Seeds python/numpy/torch (ref general.py:74-81; the cudnn knobs have no ROCm meaning here)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def iter_average(input_iter):
    input_iter = list(input_iter)
    return sum(input_iter) / len(input_iter)


def dict_merge(dct: Dict[str, Any], merge_dct: Dict[str, Any], re=True):
    """Recursive in-place merge of ``merge_dct`` into ``dct`` (ref general.py:303-330)."""
    for k, v in merge_dct.items():
        if k in dct and isinstance(dct[k], dict) and isinstance(v, collections.abc.Mapping):
            dict_merge(dct[k], v, re=False)
        else:
            dct[k] = v
    if re:
        return dct


def flatten_dict(d, parent_key="", sep="_"):
    items = []
    for k, v in d.items():
        key = parent_key + sep + k if parent_key else k
        if isinstance(v, collections.abc.MutableMapping):
            items.extend(flatten_dict(v, key, sep=sep).items())
        else:
            items.append((key, v))
    return dict(items)


def nice_dict(input_dict) -> str:
    flat = flatten_dict(input_dict, sep="") if any(isinstance(v, dict) for v in input_dict.values()) else input_dict
    return ", ".join(f"{k}:{v:.3f}" for k, v in flat.items())


def gethash(file_path) -> str:
    try:
        return subprocess.check_output(["git", "rev-parse", "HEAD"], cwd=os.path.dirname(os.path.abspath(file_path)),
                                       stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return "none"


def assert_list(func, iters) -> bool:
    return all(func(x) for x in iters)


def to_float(value):
    if torch.is_tensor(value):
        return float(value.item())
    return float(value)


def path2Path(path):
    from pathlib import Path
    return Path(path)


def write_yaml(dictionary: Dict, save_dir, save_name: str) -> None:
    import yaml
    from pathlib import Path
    Path(save_dir).mkdir(parents=True, exist_ok=True)
    with open(str(Path(save_dir) / save_name), "w") as f:
        yaml.dump(dictionary, f)
