"""Trainer base: epoch loop scaffolding, device move, checkpoint save/resume
(ref whl:deepclustering2/trainer/trainer.py:15-42, _trainer.py:29-32, _functional.py:39-53, _io.py:18-157).
Checkpoint = dict of every attribute's ``state_dict()`` + ``_buffers`` {_best_score,_start_epoch,_cur_epoch},
written to ``<RUN_PATH>/<save_dir>/{last,best}.pth`` -- the reference's format."""
from abc import ABCMeta, abstractmethod
from copy import deepcopy
from pathlib import Path

import torch

from deepclustering2.meters2 import Storage
from deepclustering2.utils import write_yaml
from deepclustering2.writer import SummaryWriter

_BUFFER_NAMES = ("_best_score", "_start_epoch", "_cur_epoch")


def _rank() -> int:
    """Rank in the data-parallel job (0 without torch.distributed)."""
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class _NullWriter:
    """What the non-writing ranks of a data-parallel job log to."""

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def __getattr__(self, name):
        if not name.startswith(("add_", "close", "flush")):      # in particular no state_dict: it must not show up in checkpoints
            raise AttributeError(name)
        return lambda *a, **k: None


class Trainer(metaclass=ABCMeta):
    RUN_PATH = str(Path.cwd() / "runs")

    def __init__(self, model, save_dir: str = "base", max_epoch: int = 100, num_batches: int = 100, device: str = "cpu",
                 configuration=None):
        assert isinstance(save_dir, str), save_dir
        if not Path(save_dir).is_absolute():
            save_dir = str(Path(self.RUN_PATH) / save_dir)
        self._save_dir = save_dir
        if self.is_writer:
            Path(self._save_dir).mkdir(exist_ok=True, parents=True)
        self._best_score, self._start_epoch, self._cur_epoch = -1, 0, 0
        self._max_epoch, self._num_batches = max_epoch, num_batches
        self._config = deepcopy(configuration)
        if self._config and self.is_writer:
            write_yaml(self._config, save_dir, save_name="config.yaml")
        self._storage = Storage()
        self._model = model
        self._device = torch.device(device)

    @property
    def is_writer(self) -> bool:
        """One process per GPU: only rank 0 touches the run directory (config.yaml, checkpoints, csv, TensorBoard)."""
        return _rank() == 0

    # ---- loop scaffolding
    def _writer_context(self):
        return SummaryWriter(str(self._save_dir)) if self.is_writer else _NullWriter()

    def start_training(self, *args, **kwargs):
        self.to(self._device)
        with self._writer_context() as self._writer:
            return self._start_training(*args, **kwargs)

    @abstractmethod
    def _start_training(self, *args, **kwargs):
        pass

    def run_epoch(self, *args, **kwargs):
        return self._run_epoch(*args, **kwargs)

    def eval_epoch(self, *args, **kwargs):
        return self._eval_epoch(*args, **kwargs)

    @abstractmethod
    def _run_epoch(self, *args, **kwargs):
        pass

    @abstractmethod
    def _eval_epoch(self, *args, **kwargs):
        pass

    def to(self, device):
        """Move every attribute that has ``.to`` (model, projectors, criteria) -- ref _functional.py:39-53."""
        for name, module in self.__dict__.items():
            if isinstance(module, torch.optim.Optimizer):
                for st in module.state.values():
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            st[k] = v.to(device)
            elif hasattr(module, "to") and callable(module.to) and not isinstance(module, torch.device):
                try:
                    module.to(device)
                except Exception:
                    pass

    # ---- checkpoints
    def state_dict(self) -> dict:
        out = {}
        for name, module in self.__dict__.items():
            if hasattr(module, "state_dict") and callable(module.state_dict):
                out[name] = module.state_dict()
        out["_buffers"] = {k: getattr(self, k) for k in _BUFFER_NAMES}
        return out

    def load_state_dict(self, state_dict: dict, strict=True) -> None:
        errors = []
        for k, v in state_dict.get("_buffers", {}).items():
            setattr(self, k, v)
        for name, module in self.__dict__.items():
            if hasattr(module, "load_state_dict") and callable(module.load_state_dict):
                try:
                    module.load_state_dict(state_dict[name])
                except KeyError:
                    pass
                except Exception as ex:  # noqa
                    errors.append(f"while copying {name} parameters, error {ex} occurs")
        if errors:
            msg = "Error(s) in loading state_dict for {}:\n\t{}".format(self.__class__.__name__, "\n\t".join(errors))
            if strict:
                raise RuntimeError(msg)
            import warnings
            warnings.warn(RuntimeWarning(msg))
        if self._cur_epoch > self._start_epoch:
            self._start_epoch = self._cur_epoch + 1

    def load_state_dict_from_path(self, path, *args, **kwargs) -> None:
        path = Path(path)
        assert path.exists(), path
        if path.is_dir() and (path / "last.pth").exists():
            path = path / "last.pth"
        elif not (path.is_file() and path.suffix in (".pth", ".pt")):
            raise FileNotFoundError(path)
        self.load_state_dict(torch.load(str(path), map_location="cpu", weights_only=False), *args, **kwargs)

    def _save_to(self, save_name, path=None):
        assert Path(save_name).suffix in (".pth", ".pt")
        path = Path(path or self._save_dir)
        path.mkdir(parents=True, exist_ok=True)
        torch.save(self.state_dict(), str(path / save_name))

    def resume_from_checkpoint(self, checkpoint, **kwargs):
        self.load_state_dict_from_path(checkpoint, **kwargs)

    def save(self, current_score: float, path=None):
        if not self.is_writer:
            self._best_score = max(self._best_score, current_score)
            return
        self._save_to("last.pth", path)
        if self._best_score < current_score:
            self._best_score = current_score
            self._save_to("best.pth", path)
