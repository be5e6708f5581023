from .warmup_scheduler import GradualWarmupScheduler  # noqa: F401
