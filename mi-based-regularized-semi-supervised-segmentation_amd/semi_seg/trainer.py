"""Trainers of the semi-supervised segmentation path (ref: semi_seg/trainer.py:24-214).

``trainer_zoos = {partial, uda, iic, udaiic}`` with the reference's constructor signature, config keys
(``config/semi.yaml``), epoch loop (train -> eval(val) -> eval(test) -> scheduler step -> storage / TensorBoard ->
save last/best -> csv) and checkpoint layout.  ``Optim.name: Adam`` resolves to the fused HIP Adam.
Data-parallel runs (one process per GPU, RCCL) wrap the optimiser's flat gradient buffer with
``miseg_amd.ddp.GradReducer`` -- see ``miseg_amd.ddp`` -- without changing anything here.
"""
import os
from copy import deepcopy
from itertools import chain
from pathlib import Path
from typing import Tuple

import torch
from torch import nn

from contrastyou import PROJECT_PATH
from deepclustering2 import optim
from deepclustering2.loss import KL_div
from deepclustering2.meters2 import EpochResultDict, StorageIncomeDict
from deepclustering2.schedulers import GradualWarmupScheduler
from deepclustering2.trainer import Trainer
from deepclustering2.type import T_loader, T_loss
from semi_seg._utils import IICLossWrapper, ProjectorWrapper
from semi_seg.epocher import (EvalEpocher, IICTrainEpocher, InferenceEpocher, TrainEpocher, UDAIICEpocher, UDATrainEpocher)

__all__ = ["trainer_zoos"]


class SemiTrainer(Trainer):
    RUN_PATH = str(Path(PROJECT_PATH) / "semi_seg" / "runs")  # noqa
    feature_positions = ["Up_conv4", "Up_conv3"]

    def __init__(self, *, model: nn.Module, labeled_loader: T_loader, unlabeled_loader: T_loader, val_loader: T_loader,
                 test_loader: T_loader, sup_criterion: T_loss, save_dir: str = "base", max_epoch: int = 100,
                 num_batches: int = 100, device: str = "cpu", configuration=None, **kwargs):
        super().__init__(model, save_dir, max_epoch, num_batches, device, configuration)
        self._labeled_loader, self._unlabeled_loader = labeled_loader, unlabeled_loader
        self._val_loader, self._test_loader = val_loader, test_loader
        self._sup_criterion = sup_criterion
        self._grad_reducer = None  # miseg_amd.ddp.GradReducer for multi-GPU runs

    def init(self):
        self._init()
        self._init_optimizer()
        self._init_scheduler(self._optimizer)

    def _init(self):
        self.set_feature_positions(self._config["Trainer"]["feature_names"])
        importance = self._config["Trainer"]["feature_importance"]
        assert isinstance(importance, list), type(importance)
        importance = [float(x) for x in importance]
        self._feature_importance = [x / sum(importance) for x in importance]
        assert len(self._feature_importance) == len(self.feature_positions)

    def _init_scheduler(self, optimizer):
        sched = self._config.get("Scheduler", None)
        if sched is None:
            return
        cosine = torch.optim.lr_scheduler.CosineAnnealingLR(
            self._optimizer, T_max=self._config["Trainer"]["max_epoch"] - sched["warmup_max"], eta_min=1e-7)
        self._scheduler = GradualWarmupScheduler(optimizer, sched["multiplier"], total_epoch=sched["warmup_max"],
                                                 after_scheduler=cosine)

    def _trainable(self):
        return self._model.parameters()

    def _init_optimizer(self):
        cfg = self._config["Optim"]
        self._optimizer = optim.__dict__[cfg["name"]](params=self._trainable(), **{k: v for k, v in cfg.items() if k != "name"})

    def _epocher_common(self):
        return dict(num_batches=self._num_batches, cur_epoch=self._cur_epoch, device=self._device,
                    feature_position=self.feature_positions, feature_importance=self._feature_importance)

    def _launch(self, epocher) -> EpochResultDict:
        epocher._reducer = self._grad_reducer
        return epocher.run()

    def _run_epoch(self, *args, **kwargs) -> EpochResultDict:
        return self._launch(TrainEpocher(self._model, self._optimizer, self._labeled_loader, self._unlabeled_loader,
                                         self._sup_criterion, 0, **self._epocher_common()))

    def _eval_epoch(self, *, loader: T_loader, **kwargs) -> Tuple[EpochResultDict, float]:
        evaler = EvalEpocher(self._model, val_loader=loader, sup_criterion=self._sup_criterion, cur_epoch=self._cur_epoch,
                             device=self._device)
        return evaler.run()

    def _start_training(self):
        for self._cur_epoch in range(self._start_epoch, self._max_epoch):
            train_result = self.run_epoch()
            with torch.no_grad():
                eval_result, cur_score = self.eval_epoch(loader=self._val_loader)
                test_result, _ = self.eval_epoch(loader=self._test_loader)
            if hasattr(self, "_scheduler"):
                self._scheduler.step()
            storage_per_epoch = StorageIncomeDict(tra=train_result, val=eval_result, test=test_result)
            self._storage.put_from_dict(storage_per_epoch, self._cur_epoch)
            self._writer.add_scalar_with_StorageDict(storage_per_epoch, self._cur_epoch)
            self.save(cur_score)
            self._storage.to_csv(self._save_dir)

    def inference(self, checkpoint=None):  # noqa
        if checkpoint is None:
            self.load_state_dict_from_path(os.path.join(self._save_dir, "best.pth"), strict=True)
        else:
            checkpoint = Path(checkpoint)
            if checkpoint.is_file():
                if checkpoint.suffix != ".pth":
                    raise FileNotFoundError(checkpoint)
            else:
                assert checkpoint.exists()
                checkpoint = checkpoint / "best.pth"
            self.load_state_dict_from_path(str(checkpoint), strict=True)
        evaler = InferenceEpocher(self._model, val_loader=self._test_loader, sup_criterion=self._sup_criterion,
                                  cur_epoch=self._cur_epoch, device=self._device)
        evaler.set_save_dir(self._save_dir)
        result, cur_score = evaler.run()
        return result, cur_score

    @classmethod
    def set_feature_positions(cls, feature_positions):
        cls.feature_positions = feature_positions


class UDATrainer(SemiTrainer):
    def _init(self):
        super()._init()
        cfg = deepcopy(self._config["UDARegCriterion"])
        self._reg_criterion = {"mse": nn.MSELoss(), "kl": KL_div()}[cfg["name"]]
        self._reg_weight = float(cfg["weight"])

    def _run_epoch(self, *args, **kwargs) -> EpochResultDict:
        return self._launch(UDATrainEpocher(self._model, self._optimizer, self._labeled_loader, self._unlabeled_loader,
                                            self._sup_criterion, reg_weight=self._reg_weight, reg_criterion=self._reg_criterion,
                                            **self._epocher_common()))


class IICTrainer(SemiTrainer):
    def _init(self):
        super()._init()
        cfg = deepcopy(self._config["IICRegParameters"])
        self._projector_wrappers = ProjectorWrapper()
        self._projector_wrappers.init_encoder(feature_names=self.feature_positions, **cfg["EncoderParams"])
        self._projector_wrappers.init_decoder(feature_names=self.feature_positions, **cfg["DecoderParams"])
        self._IIDSegWrapper = IICLossWrapper(feature_names=self.feature_positions, **cfg["LossParams"])
        self._reg_weight = float(cfg["weight"])

    def _trainable(self):
        return chain(self._model.parameters(), self._projector_wrappers.parameters())

    def _run_epoch(self, *args, **kwargs) -> EpochResultDict:
        return self._launch(IICTrainEpocher(self._model, self._projector_wrappers, self._optimizer, self._labeled_loader,
                                            self._unlabeled_loader, self._sup_criterion, reg_weight=self._reg_weight,
                                            IIDSegCriterionWrapper=self._IIDSegWrapper, **self._epocher_common()))


class UDAIICTrainer(IICTrainer):
    def _init(self):
        super()._init()
        self._iic_weight = deepcopy(self._reg_weight)
        self._reg_weight = 1.0
        cfg = deepcopy(self._config["UDARegCriterion"])
        self._reg_criterion = {"mse": nn.MSELoss(), "kl": KL_div()}[cfg["name"]]
        self._uda_weight = float(cfg["weight"])

    def _run_epoch(self, *args, **kwargs) -> EpochResultDict:
        return self._launch(UDAIICEpocher(self._model, self._projector_wrappers, self._optimizer, self._labeled_loader,
                                          self._unlabeled_loader, self._sup_criterion, self._reg_criterion, self._IIDSegWrapper,
                                          cons_weight=self._uda_weight, iic_weight=self._iic_weight, **self._epocher_common()))


trainer_zoos = {"partial": SemiTrainer, "uda": UDATrainer, "iic": IICTrainer, "udaiic": UDAIICTrainer}
