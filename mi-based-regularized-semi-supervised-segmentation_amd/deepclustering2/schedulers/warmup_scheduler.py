"""Linear warm-up wrapped around another scheduler (ref whl:deepclustering2/schedulers/warmup_scheduler.py:8-75).
Same composition as the reference -- a torch ``_LRScheduler`` that hands over to ``after_scheduler`` -- so the
lr sequence (including its hand-over quirk) is whatever torch's CosineAnnealingLR gives under this wrapper."""
from torch.optim.lr_scheduler import _LRScheduler


class GradualWarmupScheduler(_LRScheduler):
    def __init__(self, optimizer, multiplier, total_epoch, after_scheduler=None):
        if multiplier <= 1.0:
            raise ValueError("multiplier should be greater than 1.")
        self.multiplier, self.total_epoch, self.after_scheduler, self.finished = multiplier, total_epoch, after_scheduler, False
        super().__init__(optimizer)

    def get_lr(self):
        if self.last_epoch > self.total_epoch:
            if self.after_scheduler:
                if not self.finished:
                    self.after_scheduler.base_lrs = [b * self.multiplier for b in self.base_lrs]
                    self.finished = True
                return self.after_scheduler.get_lr()
            return [b * self.multiplier for b in self.base_lrs]
        return [b * ((self.multiplier - 1.0) * self.last_epoch / self.total_epoch + 1.0) for b in self.base_lrs]

    def step(self, epoch=None, metrics=None):
        if self.finished and self.after_scheduler:
            self.after_scheduler.step(None if epoch is None else epoch - self.total_epoch)
        else:
            return super().step(epoch)
