"""Type aliases and ``to_device`` (ref whl:deepclustering2/type/typecheckconvert.py:299-322)."""
from typing import Any, Iterable, Union

import torch
from torch import Tensor, nn, optim
from torch.utils.data import DataLoader

T_loader = Union[DataLoader, Iterable]
T_loss = nn.Module
T_optim = optim.Optimizer
T_iter = Iterable


def to_device(obj: Any, device, non_blocking=True):
    if torch.is_tensor(obj):
        return obj.to(device, non_blocking=non_blocking)
    if isinstance(obj, dict):
        return {k: to_device(v, device, non_blocking) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_device(v, device, non_blocking) for v in obj) if not isinstance(obj, list) else \
            [to_device(v, device, non_blocking) for v in obj]
    return obj


def to_float(value):
    return float(value.item()) if torch.is_tensor(value) else float(value)
