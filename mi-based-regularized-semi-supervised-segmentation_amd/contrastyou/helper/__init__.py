from .utils import average_iter, multiply_iter, weighted_average_iter  # noqa: F401
