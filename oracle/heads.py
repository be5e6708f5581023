"""Oracle: cluster heads (global and local) as pure functions over state_dict-style weights.

Restates ``contrastyou/trainer/_utils.py``: ClusterHead :96-134 (avg-pool -> flatten ->
Linear(C->K) [or Linear(C->128)->LeakyReLU(0.01)->Linear(128->K)] -> [L2 normalise] ->
softmax(x/T)) and LocalClusterHead :137-168 (1x1 conv(s) -> [normalise] -> channel
softmax(x/T)); SoftmaxWithT :15-23 divides by T before the softmax.
Keys follow the reference module tree: ``_headers.<s>.<idx>.{weight,bias}``.
Test infrastructure only.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor


def init_cluster_head(input_dim: int, num_clusters: int, num_subheads: int, head_type: str = "linear",
                      seed: int = 0, dtype=torch.float32) -> "OrderedDict[str, Tensor]":
    rs = np.random.RandomState(seed)

    def randn(*shape):
        return torch.from_numpy(rs.standard_normal(shape))

    sd: "OrderedDict[str, Tensor]" = OrderedDict()
    for s in range(num_subheads):
        if head_type == "linear":
            sd[f"_headers.{s}.2.weight"] = (randn(num_clusters, input_dim) / input_dim ** 0.5).to(dtype)
            sd[f"_headers.{s}.2.bias"] = (0.1 * randn(num_clusters)).to(dtype)
        else:
            sd[f"_headers.{s}.2.weight"] = (randn(128, input_dim) / input_dim ** 0.5).to(dtype)
            sd[f"_headers.{s}.2.bias"] = (0.1 * randn(128)).to(dtype)
            sd[f"_headers.{s}.4.weight"] = (randn(num_clusters, 128) / 128 ** 0.5).to(dtype)
            sd[f"_headers.{s}.4.bias"] = (0.1 * randn(num_clusters)).to(dtype)
    return sd


def init_local_cluster_head(input_dim: int, num_clusters: int, num_subheads: int, head_type: str = "linear",
                            interm_dim: int = 64, seed: int = 0, dtype=torch.float32) -> "OrderedDict[str, Tensor]":
    rs = np.random.RandomState(seed)

    def randn(*shape):
        return torch.from_numpy(rs.standard_normal(shape))

    sd: "OrderedDict[str, Tensor]" = OrderedDict()
    for s in range(num_subheads):
        if head_type == "linear":
            sd[f"_headers.{s}.0.weight"] = (randn(num_clusters, input_dim, 1, 1) / input_dim ** 0.5).to(dtype)
            sd[f"_headers.{s}.0.bias"] = (0.1 * randn(num_clusters)).to(dtype)
        else:
            sd[f"_headers.{s}.0.weight"] = (randn(interm_dim, input_dim, 1, 1) / input_dim ** 0.5).to(dtype)
            sd[f"_headers.{s}.0.bias"] = (0.1 * randn(interm_dim)).to(dtype)
            sd[f"_headers.{s}.2.weight"] = (randn(num_clusters, interm_dim, 1, 1) / interm_dim ** 0.5).to(dtype)
            sd[f"_headers.{s}.2.bias"] = (0.1 * randn(num_clusters)).to(dtype)
    return sd


def _num_subheads(sd) -> int:
    return 1 + max(int(k.split(".")[1]) for k in sd)


def cluster_head(sd, features: Tensor, temperature: float = 1.0, normalize: bool = False) -> list[Tensor]:
    """ClusterHead.forward (_utils.py:133-134): list of [N,K] simplexes, one per sub-head."""
    pooled = features.mean(dim=(2, 3))  # AdaptiveAvgPool2d((1,1)) + Flatten
    outs = []
    for s in range(_num_subheads(sd)):
        z = F.linear(pooled, sd[f"_headers.{s}.2.weight"], sd[f"_headers.{s}.2.bias"])
        if f"_headers.{s}.4.weight" in sd:
            z = F.linear(F.leaky_relu(z, 0.01), sd[f"_headers.{s}.4.weight"], sd[f"_headers.{s}.4.bias"])
        if normalize:
            z = F.normalize(z, p=2, dim=1)
        outs.append(torch.softmax(z / temperature, dim=1))
    return outs


def local_cluster_head(sd, features: Tensor, temperature: float = 1.0, normalize: bool = False) -> list[Tensor]:
    """LocalClusterHead.forward (_utils.py:167-168): list of [N,K,H,W] per-pixel simplexes."""
    outs = []
    for s in range(_num_subheads(sd)):
        z = F.conv2d(features, sd[f"_headers.{s}.0.weight"], sd[f"_headers.{s}.0.bias"])
        if f"_headers.{s}.2.weight" in sd:
            z = F.conv2d(F.leaky_relu(z, 0.01), sd[f"_headers.{s}.2.weight"], sd[f"_headers.{s}.2.bias"])
        if normalize:
            z = F.normalize(z, p=2, dim=1)
        outs.append(torch.softmax(z / temperature, dim=1))
    return outs
