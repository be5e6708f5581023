"""``optim.__dict__[name]`` lookup used by the trainers (ref whl:deepclustering2/optim/__init__.py:1-11,
semi_seg/trainer.py:67-72).  ``Adam`` resolves to the fused single-launch HIP Adam over one flat
parameter buffer (torch.optim.Adam semantics incl. L2 weight decay); every other name is torch.optim's."""
from typing import List

from torch.optim import *  # noqa: F401,F403
from torch.optim import Optimizer

from miseg_amd.flat import FusedAdam as Adam  # noqa: F401


def get_lrs_from_optimizer(optimizer: Optimizer) -> List[float]:
    return [p["lr"] for p in optimizer.param_groups]
