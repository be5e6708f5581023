"""Epoch history of a run -> ``storage.csv`` and the ``_storage`` entry of the checkpoints
(ref whl:deepclustering2/meters2/storage_interface.py:17-113).

One ``HistoricalContainer`` per ``<tra|val|test>_<meter>`` name; the csv has one row per epoch and one column per
``<name>_<key>``, in the order names and keys were first recorded -- what the reference's chain of pandas merges
produces (header pinned by tests/golden/trainer_io.npz).  ``state_dict()`` is, like the reference's, the
``defaultdict(HistoricalContainer)`` itself, so checkpoints are interchangeable; ``load_state_dict`` also accepts the plain
``{name: {epoch: record}}`` form round 1 of this repo wrote."""
from collections import defaultdict
from pathlib import Path

import pandas as pd

from .historicalContainer import HistoricalContainer


class StorageIncomeDict:
    def __init__(self, **kwargs) -> None:
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __repr__(self):
        return "\n".join(f"{k}:\n{v}" for k, v in self.__dict__.items())


class Storage:
    def __init__(self, csv_save_dir=None, csv_name="storage.csv") -> None:
        self._storage = defaultdict(HistoricalContainer)
        self._csv_save_dir, self._csv_name = csv_save_dir, csv_name

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return None

    def put(self, name, value, epoch=None, prefix="", postfix=""):
        self._storage[prefix + name + postfix].add(dict(value), epoch)

    def put_all(self, result_name, epoch_result=None, epoch=None):
        assert isinstance(result_name, str), result_name
        for k, v in (epoch_result or {}).items():
            self.put(result_name + "_" + k, v, epoch)

    def put_from_dict(self, income_dict: StorageIncomeDict, epoch: int = None):
        for k, v in income_dict.__dict__.items():
            self.put_all(k, v, epoch)
        if self._csv_save_dir:
            self.to_csv(self._csv_save_dir, name=self._csv_name)

    def get(self, name, epoch=None):
        assert name in self._storage, name
        return self._storage[name] if epoch is None else self._storage[name][epoch]

    def summary(self) -> pd.DataFrame:
        tables = [hist.summary().add_prefix(name + "_") for name, hist in self._storage.items()]
        if not tables:
            return pd.DataFrame()
        return pd.concat(tables, axis=1, join="inner")       # inner: the epochs every meter has, as the reference's merges keep

    def to_csv(self, path, name="storage.csv"):
        path = Path(path)
        path.mkdir(exist_ok=True, parents=True)
        self.summary().to_csv(str(path / name))

    @property
    def meter_names(self):
        return list(self._storage.keys())

    @property
    def storage(self):
        return self._storage

    def state_dict(self):
        return self._storage

    def load_state_dict(self, state_dict):
        restored = defaultdict(HistoricalContainer)
        for name, hist in state_dict.items():
            if isinstance(hist, HistoricalContainer):
                restored[name] = hist
            else:                                   # {epoch: record}
                for epoch, record in hist.items():
                    restored[name].add(dict(record), epoch)
        self._storage = restored
