from ._epocher import _Epocher  # noqa: F401
