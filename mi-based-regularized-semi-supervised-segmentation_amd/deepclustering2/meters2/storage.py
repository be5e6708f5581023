"""Epoch history -> ``storage.csv`` (ref whl:deepclustering2/meters2/storage_interface.py:17-113).
Column naming ``<tra|val|test>_<meter>_<key>``, one row per epoch, as the reference's pandas merge yields."""
from collections import OrderedDict, defaultdict
from pathlib import Path

import pandas as pd


class StorageIncomeDict:
    def __init__(self, **kwargs) -> None:
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __repr__(self):
        return "\n".join(f"{k}:\n{v}" for k, v in self.__dict__.items())


class Storage:
    def __init__(self, csv_save_dir=None, csv_name="storage.csv") -> None:
        self._storage = defaultdict(OrderedDict)  # name -> {epoch: {key: value}}
        self._csv_save_dir, self._csv_name = csv_save_dir, csv_name

    def put(self, name, value, epoch=None, prefix="", postfix=""):
        hist = self._storage[prefix + name + postfix]
        hist[len(hist) if epoch is None else epoch] = dict(value)

    def put_all(self, result_name, epoch_result=None, epoch=None):
        for k, v in (epoch_result or {}).items():
            self.put(result_name + "_" + k, v, epoch)

    def put_from_dict(self, income_dict: StorageIncomeDict, epoch: int = None):
        for k, v in income_dict.__dict__.items():
            self.put_all(k, v, epoch)
        if self._csv_save_dir:
            self.to_csv(self._csv_save_dir, name=self._csv_name)

    def get(self, name, epoch=None):
        return self._storage[name] if epoch is None else self._storage[name][epoch]

    def summary(self) -> pd.DataFrame:
        cols = {}
        for name, hist in self._storage.items():
            for epoch, rec in hist.items():
                for key, val in rec.items():
                    cols.setdefault(f"{name}_{key}", {})[epoch] = val
        return pd.DataFrame(cols)

    def to_csv(self, path, name="storage.csv"):
        path = Path(path)
        path.mkdir(exist_ok=True, parents=True)
        self.summary().to_csv(str(path / name))

    @property
    def meter_names(self):
        return list(self._storage.keys())

    def state_dict(self):
        return {k: dict(v) for k, v in self._storage.items()}

    def load_state_dict(self, state_dict):
        self._storage = defaultdict(OrderedDict, {k: OrderedDict(v) for k, v in state_dict.items()})
