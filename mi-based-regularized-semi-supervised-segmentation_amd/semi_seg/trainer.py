"""The four semi-supervised trainers behind ``Trainer.name`` (ref ``semi_seg/trainer.py:24-214``).

Drop-in surface: ``trainer_zoos = {partial, uda, iic, udaiic}``, the keyword-only constructor, ``init()``,
``start_training()``, ``inference(checkpoint)``, ``set_feature_positions`` and the attribute names the checkpoint tree is
keyed by (``_model``, ``_optimizer``, ``_scheduler``, ``_projector_wrappers``, ``_IIDSegWrapper``, ``_storage`` ...; the
tree itself is pinned by ``tests/golden/trainer_io.npz``).  Config sections are the ones of ``config/semi.yaml``.

Own structure: a trainer is (a) the config sections it reads in ``_init`` and (b) ONE ``_make_epocher`` that builds the
epoch object; the loop, evaluation, checkpointing and logging live once in ``SemiTrainer``.  Data-parallel runs (one
process per GPU, RCCL over xGMI) are part of the loop, not bolted on: ``attach_data_parallel`` gives the optimiser's flat
gradient buffer a ``miseg_amd.ddp.GradReducer``, every epocher is handed that reducer, every rank evaluates (identical
weights, identical scores) and only rank 0 writes ``last.pth`` / ``best.pth`` / ``storage.csv`` / TensorBoard.
"""
from __future__ import annotations

from itertools import chain
from pathlib import Path
from typing import Optional, Tuple

import torch
from torch import nn

from contrastyou import PROJECT_PATH
from deepclustering2 import optim
from deepclustering2.loss import KL_div
from deepclustering2.meters2 import EpochResultDict, StorageIncomeDict
from deepclustering2.schedulers import GradualWarmupScheduler
from deepclustering2.trainer import Trainer
from deepclustering2.type import T_loader, T_loss
from semi_seg import epocher as E
from semi_seg._utils import IICLossWrapper, ProjectorWrapper

__all__ = ["trainer_zoos"]

_CONSISTENCY = {"mse": nn.MSELoss, "kl": KL_div}   # UDARegCriterion.name (config/semi.yaml:31-33)
_COSINE_FLOOR = 1e-7                               # eta_min of the cosine phase (ref trainer.py:57-60)


def _consistency_section(section: dict) -> Tuple[nn.Module, float]:
    return _CONSISTENCY[section["name"]](), float(section["weight"])


def _checkpoint_file(checkpoint, default_dir) -> Path:
    """``None`` -> ``<save_dir>/best.pth``; a directory -> its ``best.pth``; a file must be a ``.pth``."""
    if checkpoint is None:
        return Path(default_dir) / "best.pth"
    path = Path(checkpoint)
    if path.is_dir():
        return path / "best.pth"
    if path.is_file() and path.suffix == ".pth":
        return path
    raise FileNotFoundError(path)


class SemiTrainer(Trainer):
    """``partial``: supervised KL on the labeled batch only (the unlabeled loader is drawn but unused)."""

    RUN_PATH = str(Path(PROJECT_PATH) / "semi_seg" / "runs")  # noqa
    feature_positions = ["Up_conv4", "Up_conv3"]

    def __init__(self, *, model: nn.Module, labeled_loader: T_loader, unlabeled_loader: T_loader, val_loader: T_loader,
                 test_loader: T_loader, sup_criterion: T_loss, save_dir: str = "base", max_epoch: int = 100,
                 num_batches: int = 100, device: str = "cpu", configuration=None, **kwargs):
        super().__init__(model, save_dir, max_epoch, num_batches, device, configuration)
        self._labeled_loader, self._unlabeled_loader = labeled_loader, unlabeled_loader
        self._val_loader, self._test_loader = val_loader, test_loader
        self._sup_criterion = sup_criterion
        self._grad_reducer = None

    # ------------------------------------------------------------------ set-up
    def init(self) -> None:
        self._init()
        self._init_optimizer()
        self._init_scheduler(self._optimizer)

    def _init(self) -> None:
        section = self._config["Trainer"]
        # arithmetic of the local-MI contraction: Arch.mi_precision, else by Arch.compute_dtype (f16f8 for bfloat16 / float16)
        from miseg_amd import ops
        net = getattr(self._model, "module", self._model)
        ops.set_mi_precision(ops.resolve_mi_precision(getattr(net, "compute_dtype", None), getattr(net, "mi_precision", None)))
        self.set_feature_positions(section["feature_names"])
        raw = section["feature_importance"]
        assert isinstance(raw, list), type(raw)
        weights = [float(v) for v in raw]
        assert len(weights) == len(self.feature_positions), (weights, self.feature_positions)
        total = sum(weights)
        self._feature_importance = [v / total for v in weights]

    def _trainable(self):
        return self._model.parameters()

    def _init_optimizer(self) -> None:
        section = dict(self._config["Optim"])
        factory = optim.__dict__[section.pop("name")]        # ``Adam`` resolves to the fused flat-buffer HIP Adam
        self._optimizer = factory(params=self._trainable(), **section)

    def _init_scheduler(self, optimizer) -> None:
        section = self._config.get("Scheduler")
        if section is None:
            return
        warm = section["warmup_max"]
        cosine = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=self._config["Trainer"]["max_epoch"] - warm,
                                                            eta_min=_COSINE_FLOOR)
        self._scheduler = GradualWarmupScheduler(optimizer, section["multiplier"], total_epoch=warm, after_scheduler=cosine)

    @classmethod
    def set_feature_positions(cls, feature_positions) -> None:
        cls.feature_positions = feature_positions           # class-wide, as in the reference (trainer.py:127-129)

    def attach_data_parallel(self, num_buckets: int = 3):
        """One process per GPU: bucketed RCCL all-reduce of the flat gradient, launched from autograd hooks
        (``miseg_amd.ddp``).  No-op (returns None) when torch.distributed is not initialised or has a single rank."""
        from miseg_amd import ddp
        return ddp.attach(self, num_buckets=num_buckets)

    # ------------------------------------------------------------------ one epoch
    def _epoch_args(self) -> dict:
        return dict(num_batches=self._num_batches, cur_epoch=self._cur_epoch, device=self._device,
                    feature_position=self.feature_positions, feature_importance=self._feature_importance)

    def _make_epocher(self):
        return E.TrainEpocher(self._model, self._optimizer, self._labeled_loader, self._unlabeled_loader, self._sup_criterion, 0,
                              **self._epoch_args())

    def _run_epoch(self, *args, **kwargs) -> EpochResultDict:
        epocher = self._make_epocher()
        epocher._reducer = self._grad_reducer
        return epocher.run()

    def _eval_epoch(self, *, loader: T_loader, **kwargs) -> Tuple[EpochResultDict, float]:
        return E.EvalEpocher(self._model, val_loader=loader, sup_criterion=self._sup_criterion, cur_epoch=self._cur_epoch,
                             device=self._device).run()

    # ------------------------------------------------------------------ the loop
    def _start_training(self) -> None:
        for epoch in range(self._start_epoch, self._max_epoch):
            self._cur_epoch = epoch
            trained = self.run_epoch()
            with torch.no_grad():
                validated, score = self.eval_epoch(loader=self._val_loader)
                tested, _ = self.eval_epoch(loader=self._test_loader)
            scheduler = getattr(self, "_scheduler", None)
            if scheduler is not None:
                scheduler.step()
            self._log_epoch(StorageIncomeDict(tra=trained, val=validated, test=tested), score)

    def _log_epoch(self, record: StorageIncomeDict, score: float) -> None:
        """History, TensorBoard, ``last.pth`` / ``best.pth``, ``storage.csv`` -- the writing rank only.  Every rank tracks
        ``_best_score`` so that all ranks agree on it if the writer role ever moves."""
        self._storage.put_from_dict(record, self._cur_epoch)
        if not self.is_writer:
            self._best_score = max(self._best_score, score)
            return
        self._writer.add_scalar_with_StorageDict(record, self._cur_epoch)
        self.save(score)
        self._storage.to_csv(self._save_dir)

    # ------------------------------------------------------------------ inference
    def inference(self, checkpoint=None):  # noqa
        if checkpoint is not None and not Path(checkpoint).exists():
            raise AssertionError(checkpoint)         # the reference asserts the path (semi_seg/trainer.py:112-115) before resolving it
        target = _checkpoint_file(checkpoint, self._save_dir)
        self.load_state_dict_from_path(str(target), strict=True)
        runner = E.InferenceEpocher(self._model, val_loader=self._test_loader, sup_criterion=self._sup_criterion,
                                    cur_epoch=self._cur_epoch, device=self._device)
        runner.set_save_dir(self._save_dir)
        return runner.run()


class UDATrainer(SemiTrainer):
    """``uda``: + ``weight`` x consistency between f(flip(x)) and flip(f(x))."""

    def _init(self) -> None:
        super()._init()
        self._reg_criterion, self._reg_weight = _consistency_section(self._config["UDARegCriterion"])

    def _make_epocher(self):
        return E.UDATrainEpocher(self._model, self._optimizer, self._labeled_loader, self._unlabeled_loader, self._sup_criterion,
                                 reg_criterion=self._reg_criterion, reg_weight=self._reg_weight, **self._epoch_args())


class IICTrainer(SemiTrainer):
    """``iic``: + ``weight`` x importance-weighted IIC mutual information over the tapped features; the projector heads
    are trained with the network."""

    def _init(self) -> None:
        super()._init()
        section = self._config["IICRegParameters"]
        heads = ProjectorWrapper()
        heads.init_encoder(feature_names=self.feature_positions, **section["EncoderParams"])
        heads.init_decoder(feature_names=self.feature_positions, **section["DecoderParams"])
        self._projector_wrappers = heads
        self._IIDSegWrapper = IICLossWrapper(feature_names=self.feature_positions, **section["LossParams"])
        self._reg_weight = float(section["weight"])

    def _trainable(self):
        return chain(self._model.parameters(), self._projector_wrappers.parameters())

    def _make_epocher(self):
        return E.IICTrainEpocher(self._model, self._projector_wrappers, self._optimizer, self._labeled_loader, self._unlabeled_loader,
                                 self._sup_criterion, IIDSegCriterionWrapper=self._IIDSegWrapper, reg_weight=self._reg_weight,
                                 **self._epoch_args())


class UDAIICTrainer(IICTrainer):
    """``udaiic``: ``UDARegCriterion.weight`` x consistency + ``IICRegParameters.weight`` x IIC; the combined
    regulariser enters the loss with weight 1 (ref trainer.py:187-196, epocher.py:294-296)."""

    def _init(self) -> None:
        super()._init()
        self._iic_weight, self._reg_weight = self._reg_weight, 1.0
        self._reg_criterion, self._uda_weight = _consistency_section(self._config["UDARegCriterion"])

    def _make_epocher(self):
        return E.UDAIICEpocher(self._model, self._projector_wrappers, self._optimizer, self._labeled_loader, self._unlabeled_loader,
                               self._sup_criterion, self._reg_criterion, self._IIDSegWrapper, cons_weight=self._uda_weight,
                               iic_weight=self._iic_weight, **self._epoch_args())


trainer_zoos = {"partial": SemiTrainer, "uda": UDATrainer, "iic": IICTrainer, "udaiic": UDAIICTrainer}
