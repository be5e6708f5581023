"""Feature taps, projector-head banks and per-tap IIC criteria of the semi-supervised trainers.

Public surface of ref ``semi_seg/_utils.py:12-224`` -- ``FeatureExtractor``, ``ProjectorWrapper`` (``init_encoder`` /
``init_decoder``), ``IICLossWrapper``, the two cluster-wrapper classes and ``IIDLoss`` -- with the reference's
constructor arguments, iteration order (encoder taps first, then decoder taps: the epocher zips taps, heads and criteria
positionally, ref ``semi_seg/epocher.py:244-246``) and checkpoint keys (``_encoder_projectors._clusters.<tap>...``,
``_decoder_projectors._clusters.<tap>...``).  Everything behind that surface is this repo's own: one ``_HeadBank``
builds either kind of head from a factory, per-tap options are broadcast by ``_per_tap``, and the extractor is a plain
name -> latest-output table whose hooks can also notify a listener the moment a tap exists (the IIC branch starts on its
own HIP stream from that callback, ``semi_seg/epocher.py::IICTrainEpocher._on_feature``).
"""
from __future__ import annotations

from typing import Callable, Dict, Iterator, List, Optional, Sequence, Union

from torch import Tensor, nn

from contrastyou.arch import UNet
from contrastyou.losses.iic_loss import IIDLoss as _GlobalIID, IIDSegmentationSmallPathLoss
from contrastyou.trainer._utils import ClusterHead, LocalClusterHead

# which side of the U-Net a tap sits on decides the head kind (pooled vs per-pixel) and the criterion (global vs local MI)
_SIDE: Dict[str, str] = {**{f"Conv{i}": "encoder" for i in range(1, 6)},
                         **{n: "decoder" for n in ("Up5", "Up_conv5", "Up4", "Up_conv4", "Up3", "Up_conv3", "Up2", "Up_conv2",
                                                   "DeConv_1x1")}}


def _names(feature_names: Union[str, Sequence[str]]) -> List[str]:
    return [feature_names] if isinstance(feature_names, str) else list(feature_names)


def _on_side(feature_names, side: str) -> List[str]:
    return [f for f in _names(feature_names) if _SIDE.get(f) == side]


def _per_tap(value, count: int, what: str) -> list:
    """A scalar option applies to every tap; a list must give one entry per tap."""
    if isinstance(value, (list, tuple)):
        if len(value) != count:
            raise AssertionError(f"{what}: {len(value)} values for {count} feature(s)")
        return list(value)
    return [value] * count


class IIDLoss(_GlobalIID):
    """The global criterion as the wrapper hands it out: the loss term only (ref _utils.py:12-15)."""

    def forward(self, x_out: Tensor, x_tf_out: Tensor):
        loss, _without_lambda, _joint = super().forward(x_out, x_tf_out)
        return loss


class FeatureExtractor(nn.Module):
    """``with FeatureExtractor(net, names) as fx: net(x); fx["Up_conv2"]`` -- latest output of each named sub-module,
    kept in ``names`` order (ref _utils.py:38-78).  ``fx.on_feature = fn`` makes every hook call ``fn(name, output)`` as
    soon as the module has produced it."""

    def __init__(self, net: UNet, feature_names: Union[List[str], str]) -> None:
        super().__init__()
        self._net = net
        self._feature_names = _names(feature_names)
        unknown = [f for f in self._feature_names if f not in _SIDE]
        assert not unknown, f"unknown feature name(s) {unknown}"
        self.on_feature: Optional[Callable[[str, Tensor], None]] = None
        self._latest: Dict[str, Optional[Tensor]] = {}
        self._handles: list = []

    def _hook_for(self, name: str):
        def hook(_module, _inputs, output):
            self._latest[name] = output
            listener = self.on_feature
            if listener is not None:
                listener(name, output)
        return hook

    def __enter__(self) -> "FeatureExtractor":
        self._latest = {name: None for name in self._feature_names}
        self._handles = [getattr(self._net, name).register_forward_hook(self._hook_for(name)) for name in self._feature_names]
        return self

    def __exit__(self, *exc) -> None:
        while self._handles:
            self._handles.pop().remove()
        self._latest = {}

    def __getitem__(self, name: str) -> Tensor:
        return self._latest[name]

    def get_feature_from_num(self, num: int) -> Tensor:
        return self._latest[self._feature_names[num]]

    def __iter__(self) -> Iterator[Tensor]:
        return iter([self._latest[name] for name in self._feature_names])


class _HeadBank(nn.Module):
    """``_clusters[tap]`` = one multi-sub-head projector per tap; options broadcast per tap."""

    head_factory: Callable[..., nn.Module] = None

    def __init__(self, feature_names, head_types="linear", num_subheads=5, num_clusters=10, normalize=False) -> None:
        super().__init__()
        self._feature_names = _names(feature_names)
        n = len(self._feature_names)
        options = zip(self._feature_names, _per_tap(head_types, n, "head_types"), _per_tap(num_clusters, n, "num_clusters"),
                      _per_tap(num_subheads, n, "num_subheads"), _per_tap(normalize, n, "normalize"))
        self._clusters = nn.ModuleDict({
            tap: type(self).head_factory(input_dim=UNet.dimension_dict[tap], head_type=kind, num_clusters=clusters,
                                         num_subheads=subheads, normalize=norm)
            for tap, kind, clusters, subheads, norm in options})

    def __len__(self) -> int:
        return len(self._clusters)

    def __iter__(self):
        return iter(self._clusters.values())

    def __getitem__(self, tap: str) -> nn.Module:
        return self._clusters[tap]

    def __contains__(self, tap: str) -> bool:
        return tap in self._clusters


class LocalClusterWrappaer(_HeadBank):      # (sic) the reference's public name, ref _utils.py:81
    head_factory = LocalClusterHead


class EncoderClusterWrapper(_HeadBank):
    head_factory = ClusterHead


class ProjectorWrapper(nn.Module):
    """Encoder bank + decoder bank; iterating yields the heads in tap order, encoder side first."""

    ENCODER_INITIALIZED = False
    DECODER_INITIALIZED = False

    def init_encoder(self, feature_names, head_types="linear", num_subheads=5, num_clusters=10, normalize=False):
        self._encoder_names = _on_side(feature_names, "encoder")
        self._encoder_projectors = EncoderClusterWrapper(self._encoder_names, head_types, num_subheads, num_clusters, normalize)
        self.ENCODER_INITIALIZED = True

    def init_decoder(self, feature_names, head_types="linear", num_subheads=5, num_clusters=10, normalize=False):
        self._decoder_names = _on_side(feature_names, "decoder")
        self._decoder_projectors = LocalClusterWrappaer(self._decoder_names, head_types, num_subheads, num_clusters, normalize)
        self.DECODER_INITIALIZED = True

    def _banks(self) -> List[_HeadBank]:
        return [getattr(self, attr) for flag, attr in ((self.ENCODER_INITIALIZED, "_encoder_projectors"),
                                                       (self.DECODER_INITIALIZED, "_decoder_projectors")) if flag]

    @property
    def feature_names(self) -> List[str]:
        return self._encoder_names + self._decoder_names

    def __getitem__(self, tap: str) -> nn.Module:
        for bank in self._banks():
            if tap in bank:
                return bank[tap]
        raise IndexError(tap)

    def __iter__(self):
        if not (self.ENCODER_INITIALIZED and self.DECODER_INITIALIZED):
            raise RuntimeError(f"Encoder_projectors or Decoder_projectors are not initialized in {self.__class__.__name__}.")
        for bank in self._banks():
            yield from bank


class IICLossWrapper(nn.Module):
    """One criterion per tap: global ``IIDLoss`` on encoder taps, patch-averaged local MI with the tap's own displacement
    range / patch size on decoder taps (ref _utils.py:178-224)."""

    def __init__(self, feature_names, paddings, patch_sizes) -> None:
        super().__init__()
        names = _names(feature_names)
        self._encoder_features, self._decoder_features = _on_side(names, "encoder"), _on_side(names, "decoder")
        assert len(names) == len(self._encoder_features) + len(self._decoder_features), names
        criteria = {tap: IIDLoss() for tap in self._encoder_features}
        n = len(self._decoder_features)
        if n:
            for tap, pad, patch in zip(self._decoder_features, _per_tap(paddings, n, "paddings"), _per_tap(patch_sizes, n, "patch_sizes")):
                criteria[tap] = IIDSegmentationSmallPathLoss(padding=pad, patch_size=patch)
        self._LossModuleDict = nn.ModuleDict(criteria)

    def __getitem__(self, tap: str) -> nn.Module:
        if tap not in self._LossModuleDict:
            raise IndexError(tap)
        return self._LossModuleDict[tap]

    def __iter__(self):
        return iter(self._LossModuleDict.values())

    def items(self):
        return self._LossModuleDict.items()

    @property
    def feature_names(self) -> List[str]:
        return self._encoder_features + self._decoder_features
