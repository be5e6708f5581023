from .general import *  # noqa: F401,F403
