"""Epoch driver base class (ref whl:deepclustering2/epoch/_epocher.py:26-101): ``run()`` moves the model to
the device, opens the meters and the progress bar, then calls ``_run()``."""
from abc import ABCMeta, abstractmethod
from contextlib import contextmanager

import torch

from deepclustering2.meters2 import MeterInterface
from deepclustering2.tqdm import tqdm


class _Epocher(metaclass=ABCMeta):
    def __init__(self, model, num_batches: int = None, cur_epoch=0, device="cpu") -> None:
        self._model, self._device, self._num_batches, self._cur_epoch = model, device, num_batches, cur_epoch

    @property
    def device(self):
        return self._device if isinstance(self._device, torch.device) else torch.device(self._device)

    @contextmanager
    def _register_indicator(self):
        assert isinstance(self._num_batches, int), self._num_batches
        indicator = tqdm(range(self._num_batches)).set_desc_from_epocher(self)
        yield indicator
        indicator._print_description()

    @contextmanager
    def _register_meters(self):
        yield self._configure_meters(MeterInterface())

    @abstractmethod
    def _configure_meters(self, meters: MeterInterface) -> MeterInterface:
        return meters

    @abstractmethod
    def _run(self, *args, **kwargs):
        pass

    def run(self, *args, **kwargs):
        self.to(self._device)
        with self._register_meters() as self.meters, self._register_indicator() as self._indicator:
            return self._run(*args, **kwargs)

    def to(self, device=torch.device("cpu")):
        device = torch.device(device) if isinstance(device, str) else device
        self._model.to(device)
        self._device = device
