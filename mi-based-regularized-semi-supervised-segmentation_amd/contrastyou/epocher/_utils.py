"""Batch unpackers used by the epochers (ref: contrastyou/epocher/_utils.py:25-33)."""


def _to(x, device, non_blocking):
    if isinstance(x, (list, tuple)):
        return [_to(v, device, non_blocking) for v in x]
    return x.to(device, non_blocking=non_blocking)


def preprocess_input_with_twice_transformation(data, device, non_blocking=True):
    [(image, target), (image_tf, target_tf)] = _to(data[0], device, non_blocking)
    return (image, target), (image_tf, target_tf), data[1], data[2], data[3]


def preprocess_input_with_single_transformation(data, device, non_blocking=True):
    return data[0][0].to(device, non_blocking=non_blocking), data[0][1].to(device, non_blocking=non_blocking), data[1], data[2], data[3]


# ---- prediction dumps of the InferenceEpocher (ref: contrastyou/epocher/_utils.py:88-118; skimage.io.imsave -> PIL)
def _write_single_png(mask, save_dir: str, filename: str):
    import os

    import numpy as np
    from PIL import Image
    assert len(mask.shape) == 2, mask.shape
    os.makedirs(save_dir, exist_ok=True)
    Image.fromarray(mask.detach().cpu().numpy().astype(np.uint8)).save(os.path.join(save_dir, filename + ".png"))


def write_predict(predict_logit, save_dir: str, filenames):
    import os
    assert len(predict_logit.shape) == 4, predict_logit.shape
    filenames = [filenames] if isinstance(filenames, str) else filenames
    assert len(filenames) == len(predict_logit)
    for m, f in zip(predict_logit.max(1)[1], filenames):
        _write_single_png(m, os.path.join(save_dir, "pred"), f)


def write_img_target(image, target, save_dir: str, filenames):
    import os
    filenames = [filenames] if isinstance(filenames, str) else filenames
    image, target = image.squeeze(1), target.squeeze(1)
    assert image.shape == target.shape
    for img, f in zip(image, filenames):
        _write_single_png(img * 255, os.path.join(save_dir, "img"), f)
    for targ, f in zip(target, filenames):
        _write_single_png(targ, os.path.join(save_dir, "gt"), f)
