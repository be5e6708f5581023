from .decorator import FixRandomSeed  # noqa: F401
