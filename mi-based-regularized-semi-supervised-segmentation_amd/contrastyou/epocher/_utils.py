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
