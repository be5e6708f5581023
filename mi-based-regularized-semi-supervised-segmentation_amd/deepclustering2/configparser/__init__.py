from .config_manager import ConfigManger  # noqa: F401
