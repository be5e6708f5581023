from .SummaryWriter import SummaryWriter  # noqa: F401
