"""List-averaging helpers used by the epochers (ref: contrastyou/helper/utils.py:46-56)."""


def average_iter(a_list):
    """Plain mean of a list of scalars/tensors (ref :46-47)."""
    return sum(a_list) / float(len(a_list))


def multiply_iter(iter_a, iter_b):
    return [a * b for a, b in zip(iter_a, iter_b)]


def weighted_average_iter(a_list, weight_list):
    """Weighted mean; the reference adds 1e-16 to the weight sum (ref :54-56) -- kept for parity."""
    denom = sum(weight_list) + 1e-16
    return sum(multiply_iter(a_list, weight_list)) / denom
