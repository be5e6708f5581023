"""The reference's transform presets (semi_seg/augment.py:7-52) as declarative recipes for the device pipeline
(miseg_amd/slices.py).  Each ``Recipe`` lists what the corresponding SequentialWrapper(Twice) is built from; the random
draws and the pixel arithmetic those objects imply are restated in slices.plan_* and csrc/augment.hip."""
from miseg_amd.slices import Recipe

_JITTER = ((0.5, 1.5), (0.5, 1.5), (0.5, 1.5))  # ColorJitter(brightness, contrast, saturation)


class ACDCStrongTransforms:
    pretrain = Recipe(geo=(("rotate", 45), ("vflip", 0.5), ("hflip", 0.5), ("random_crop", 224)), jitter=_JITTER,
                      twice=True, total_freedom=True)
    label = Recipe(geo=(("random_crop", 224), ("rotate", 30)), jitter=None, twice=True, total_freedom=True)
    val = Recipe(geo=(("center_crop", 224),), jitter=None, twice=False)
    trainval = Recipe(geo=(("random_crop", 224),), jitter=None, twice=True, total_freedom=True)
