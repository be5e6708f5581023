from .historical_container import HistoricalContainer  # noqa: F401
