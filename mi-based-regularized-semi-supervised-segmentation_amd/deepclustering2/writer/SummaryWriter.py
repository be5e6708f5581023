"""TensorBoard scalar writer with the reference's tag scheme (ref whl:deepclustering2/writer/SummaryWriter.py:15-54).
TensorBoard is optional in this image: without it the writer is a no-op context manager."""
from pathlib import Path

try:
    from torch.utils.tensorboard import SummaryWriter as _Base  # needs the tensorboard package
    _HAVE_TB = True
except Exception:  # pragma: no cover
    _Base, _HAVE_TB = object, False


class SummaryWriter(_Base):
    def __init__(self, log_dir=None, comment="", **kwargs):
        self._enabled = _HAVE_TB
        if self._enabled:
            log_dir = Path(log_dir)
            assert log_dir.exists() and log_dir.is_dir(), log_dir
            super().__init__(str(log_dir / "tensorboard"), comment, **kwargs)

    def add_scalar_with_tag(self, tag, tag_scalar_dict, global_step=None, walltime=None):
        if not self._enabled:
            return
        for k, v in tag_scalar_dict.items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    self.add_scalar(f"{tag}/{k}/{kk}", vv, global_step, walltime)
            else:
                self.add_scalar(f"{tag}/{k}", v, global_step, walltime)

    def add_scalar_with_StorageDict(self, storage_dict, epoch: int):
        for k, v in storage_dict.__dict__.items():
            self.add_scalar_with_tag(k, v, global_step=epoch)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self._enabled:
            self.close()
