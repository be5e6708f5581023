"""miseg_amd: MI355X-native engine under the reference-compatible ``contrastyou`` / ``semi_seg`` surface.

``csrc/`` holds the hand-written gfx950 kernels and the C ABI (``include/miseg_hip.h``);
``_cabi`` binds it with ctypes; ``ops`` wraps the entry points as autograd functions.
"""
from . import _cabi  # noqa: F401

__all__ = ["_cabi"]
