"""Per-meter epoch history, at the import path and with the instance attributes of the reference's class
(ref whl:deepclustering2/meters2/historicalContainer/historical_container.py:14-93).

Why the path and the two attribute names matter: the reference pickles its ``Storage`` state -- a
``defaultdict(HistoricalContainer)`` of these objects -- straight into ``last.pth`` / ``best.pth``
(whl:deepclustering2/meters2/storage_interface.py:34-35, trainer/_io.py:51-60).  Pickle stores
``module.ClassName`` plus the instance ``__dict__`` (``_record_dict``: epoch -> {key: value}, ``_current_epoch``), so a class
of that name here makes reference checkpoints load in this package and this package's checkpoints load in the reference."""
from __future__ import annotations

from collections import OrderedDict
from typing import Any, Dict, Optional

import pandas as pd

__all__ = ["HistoricalContainer"]


class HistoricalContainer:
    def __init__(self) -> None:
        self._record_dict: "OrderedDict[int, Dict[str, float]]" = OrderedDict()
        self._current_epoch: int = 0

    # ---- recording
    def add(self, input_dict: Dict[str, float], epoch: Optional[int] = None) -> None:
        """Record one epoch.  As in the reference, a falsy ``epoch`` (None -- or 0) means 'the next one'."""
        if epoch:
            self._current_epoch = epoch
        self._record_dict[self._current_epoch] = input_dict
        self._current_epoch += 1

    def reset(self) -> None:
        self._record_dict = OrderedDict()
        self._current_epoch = 0

    # ---- reading
    @property
    def record_dict(self):
        return self._record_dict

    @property
    def current_epoch(self) -> int:
        return self._current_epoch

    def get_record_dict(self, epoch=None):
        if epoch is None:
            return self._record_dict
        assert epoch in self._record_dict, f"epoch {epoch} not saved in {list(self._record_dict)}"
        return self._record_dict[epoch]

    def __getitem__(self, epoch):
        return self._record_dict[epoch]

    def summary(self) -> pd.DataFrame:
        """One row per recorded epoch, one column per key."""
        return pd.DataFrame.from_dict(self._record_dict, orient="index")

    def __repr__(self) -> str:
        return str(self.summary())

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return None

    # ---- checkpoints
    def state_dict(self) -> Dict[str, Any]:
        return self.__dict__

    def load_state_dict(self, state_dict: Dict[str, Any]) -> None:
        self.__dict__.update(state_dict)
