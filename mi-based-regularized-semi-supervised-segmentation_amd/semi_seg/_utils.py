"""Feature taps, projector heads and per-feature IIC criteria (ref: semi_seg/_utils.py:12-224).

Same classes / constructor arguments / iteration order as the reference so ``semi_seg.trainer`` wires them
identically: encoder taps first (global ``IIDLoss``), then decoder taps (``IIDSegmentationSmallPathLoss``
with per-feature padding / patch size).
"""
from __future__ import annotations

from itertools import repeat
from typing import Iterable, List, Union

from torch import Tensor, nn

from contrastyou.arch import UNet
from contrastyou.losses.iic_loss import IIDLoss as _IIDLoss, IIDSegmentationSmallPathLoss
from contrastyou.trainer._utils import ClusterHead as _EncoderClusterHead, LocalClusterHead as _LocalClusterHead

_ENCODER = ["Conv1", "Conv2", "Conv3", "Conv4", "Conv5"]
_DECODER = ["Up5", "Up_conv5", "Up4", "Up_conv4", "Up3", "Up_conv3", "Up2", "Up_conv2", "DeConv_1x1"]


class IIDLoss(_IIDLoss):
    """Returns only the loss term (ref _utils.py:12-15)."""

    def forward(self, x_out: Tensor, x_tf_out: Tensor):
        return super().forward(x_out, x_tf_out)[0]


def _filter_encodernames(feature_list):
    return [f for f in feature_list if f in _ENCODER]


def _filter_decodernames(feature_list):
    return [f for f in feature_list if f in _DECODER]


def _nlist(n):
    def parse(x):
        if isinstance(x, Iterable) and not isinstance(x, str):
            assert len(x) == n, (len(x), n)
            return list(x)
        return list(repeat(x, n))
    return parse


class FeatureExtractor(nn.Module):
    """Context manager that taps named sub-modules with forward hooks (ref _utils.py:38-78)."""

    class _Tap:
        feature = None

        def __init__(self, owner=None, name=None):
            self._owner, self._name = owner, name

        def __call__(self, _module, _inputs, result):
            self.feature = result
            cb = getattr(self._owner, "on_feature", None)   # optional: consumers that want the tap the moment it exists
            if cb is not None:
                cb(self._name, result)

    def __init__(self, net: UNet, feature_names: Union[List[str], str]) -> None:
        super().__init__()
        self._net = net
        self._feature_names = [feature_names] if isinstance(feature_names, str) else feature_names
        self.on_feature = None
        for f in self._feature_names:
            assert f in _ENCODER + _DECODER, f

    def __enter__(self):
        self._feature_exactors, self._hook_handlers = {}, {}
        for f in self._feature_names:
            tap = self._Tap(self, f)
            self._hook_handlers[f] = getattr(self._net, f).register_forward_hook(tap)
            self._feature_exactors[f] = tap
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        for handle in self._hook_handlers.values():
            handle.remove()
        del self._feature_exactors, self._hook_handlers

    def __getitem__(self, item):
        return self._feature_exactors[item].feature

    def get_feature_from_num(self, num):
        return self[self._feature_names[num]]

    def __iter__(self):
        for tap in self._feature_exactors.values():
            yield tap.feature


class LocalClusterWrappaer(nn.Module):
    """One (Local)ClusterHead per feature name (ref _utils.py:81-134)."""

    def __init__(self, feature_names, head_types="linear", num_subheads=5, num_clusters=10, normalize=False) -> None:
        super().__init__()
        self._feature_names = [feature_names] if isinstance(feature_names, str) else feature_names
        n = _nlist(len(self._feature_names))
        self._clusters = nn.ModuleDict()
        for f, h, c, s, nm in zip(self._feature_names, n(head_types), n(num_clusters), n(num_subheads), n(normalize)):
            self._clusters[f] = self._create_clusterheads(input_dim=UNet.dimension_dict[f], head_type=h, num_clusters=c,
                                                          num_subheads=s, normalize=nm)

    def __len__(self):
        return len(self._feature_names)

    def __iter__(self):
        yield from self._clusters.values()

    def __getitem__(self, item):
        return self._clusters[item]

    @staticmethod
    def _create_clusterheads(*args, **kwargs):
        return _LocalClusterHead(*args, **kwargs)


class EncoderClusterWrapper(LocalClusterWrappaer):
    @staticmethod
    def _create_clusterheads(*args, **kwargs):
        return _EncoderClusterHead(*args, **kwargs)


class ProjectorWrapper(nn.Module):
    ENCODER_INITIALIZED = False
    DECODER_INITIALIZED = False

    def init_encoder(self, feature_names, head_types="linear", num_subheads=5, num_clusters=10, normalize=False):
        self._encoder_names = _filter_encodernames(feature_names)
        self._encoder_projectors = EncoderClusterWrapper(self._encoder_names, head_types, num_subheads, num_clusters, normalize)
        self.ENCODER_INITIALIZED = True

    def init_decoder(self, feature_names, head_types="linear", num_subheads=5, num_clusters=10, normalize=False):
        self._decoder_names = _filter_decodernames(feature_names)
        self._decoder_projectors = LocalClusterWrappaer(self._decoder_names, head_types, num_subheads, num_clusters, normalize)
        self.DECODER_INITIALIZED = True

    @property
    def feature_names(self):
        return self._encoder_names + self._decoder_names

    def __getitem__(self, item):
        if self.ENCODER_INITIALIZED and item in self._encoder_projectors._feature_names:
            return self._encoder_projectors[item]
        if self.DECODER_INITIALIZED and item in self._decoder_projectors._feature_names:
            return self._decoder_projectors[item]
        raise IndexError(item)

    def __iter__(self):
        if not (self.ENCODER_INITIALIZED and self.DECODER_INITIALIZED):
            raise RuntimeError(f"Encoder_projectors or Decoder_projectors are not initialized in {self.__class__.__name__}.")
        yield from self._encoder_projectors
        yield from self._decoder_projectors


class IICLossWrapper(nn.Module):
    def __init__(self, feature_names, paddings, patch_sizes) -> None:
        super().__init__()
        feature_names = [feature_names] if isinstance(feature_names, str) else feature_names
        self._encoder_features = _filter_encodernames(feature_names)
        self._decoder_features = _filter_decodernames(feature_names)
        assert len(feature_names) == len(self._encoder_features) + len(self._decoder_features)
        self._LossModuleDict = nn.ModuleDict()
        for f in self._encoder_features:
            self._LossModuleDict[f] = IIDLoss()
        if self._decoder_features:
            n = _nlist(len(self._decoder_features))
            for f, p, size in zip(self._decoder_features, n(paddings), n(patch_sizes)):
                self._LossModuleDict[f] = IIDSegmentationSmallPathLoss(padding=p, patch_size=size)

    def __getitem__(self, item):
        if item in self._LossModuleDict.keys():
            return self._LossModuleDict[item]
        raise IndexError(item)

    def __iter__(self):
        yield from self._LossModuleDict.values()

    def items(self):
        return self._LossModuleDict.items()

    @property
    def feature_names(self):
        return self._encoder_features + self._decoder_features
