"""Train / eval epochers of the semi-supervised segmentation step on the MI355X kernels.

Drop-in for ref ``semi_seg/epocher.py``: same classes, constructor signatures and meter names
(``lr, sup_loss, reg_loss, sup_dice{DSC1..,DSC_mean}, mi, individual_mis{<feature>}, uda, iic_weight,
uda_weight``).  One step = one U-Net forward on cat[labeled, unlabeled, flip(unlabeled)], supervised KL,
regulariser, one backward, Adam (ref :137-188).  What changed underneath:

* the per-sample ``clone().flip()`` Python loops (ref :148-149, :160-161, :264-266) became index math:
  decisions are still drawn from Python's ``random`` under ``FixRandomSeed(seed)`` in the reference's
  order, packed into an int32 bit-mask tensor and applied inside the consuming kernels;
* ``softmax -> one-hot -> KL`` (ref :165-166) and ``softmax, softmax -> MSE`` (ref :221-224) are single
  fused kernels with fused backward;
* slicing / chunking / cat of tapped features and the five sub-heads (ref :258-273) are one gather-fused
  launch per tap; the IIC losses run on the MFMA joint kernels;
* the ~7 ``.item()`` host syncs per iteration (ref :182-185, :225, :278-282) are ONE device-to-host copy
  of all meter scalars.
"""
from __future__ import annotations

import os
import random
from typing import List, Optional, Tuple, Union

import torch
from torch import Tensor, nn

from contrastyou.epocher._utils import preprocess_input_with_single_transformation  # noqa
from contrastyou.epocher._utils import preprocess_input_with_twice_transformation  # noqa
from contrastyou.epocher._utils import write_img_target, write_predict
from contrastyou.helper import average_iter, weighted_average_iter
from contrastyou.trainer._utils import ClusterHead  # noqa
from deepclustering2.augment.tensor_augment import TensorRandomFlip
from deepclustering2.decorator import FixRandomSeed
from deepclustering2.epoch import _Epocher  # noqa
from deepclustering2.loss import KL_div
from deepclustering2.meters2 import (AverageValueMeter, EpochResultDict, MeterInterface, MultipleAverageValueMeter, SurfaceMeter,
                                     UniversalDice)
from deepclustering2.optim import get_lrs_from_optimizer
from deepclustering2.type import T_loader, T_loss, T_optim
from deepclustering2.utils import class2one_hot
from miseg_amd import checks, lazy, ops, stepio, unet_ops
from miseg_amd.lazy import LinearLoss
from miseg_amd.tape import keep as _keep
from semi_seg._utils import FeatureExtractor, IICLossWrapper, ProjectorWrapper

_DEBUG_ASSERTS = os.environ.get("MISEG_ASSERTS", "0") == "1"
_READBACK_EARLY = os.environ.get("MISEG_READBACK_EARLY", "1") != "0"     # Dice counts + report launched before backward (0: after it, on the step's tail)
_GUARD_STEP = os.environ.get("MISEG_GUARD_STEP", "1") != "0"   # a failed deferred check turns the iteration's Adam launch into a no-op


def _fused(fn):
    fn._miseg_fused = True
    return fn


class _num_class_mixin:
    _model: nn.Module

    @property
    def num_classes(self):
        return self._model.num_classes


class _Pending:
    """Device scalars whose host values are fetched with a single synchronising copy per iteration; the deferred
    assertion flags of miseg_amd.checks ride in the same copy and are raised right after it."""

    def __init__(self):
        self._names: List[str] = []
        self._vals: List[Tensor] = []
        self.checks: list = []
        self._static = None

    def put(self, name: str, value) -> None:
        """value: a device scalar or a symbolic miseg_amd.lazy.LinearLoss (evaluated with all others at fetch time)."""
        self._names.append(name)
        self._vals.append(value.detach() if isinstance(value, (Tensor, LinearLoss)) else value)

    def precompute(self) -> Optional[Tensor]:
        """Evaluate everything recorded so far NOW (values and check flags: ``device_values``) and keep the result for the
        iteration's ``post`` / ``fetch``; returns the check flags (a view of that vector, None without checks) -- the optimiser
        launch is guarded with them.  Nothing may be ``put`` between this call and the post."""
        if self._static is None:
            if not self._vals and not self.checks:
                return None
            scalars = self.device_values()
            self._static = (self._names, scalars, list(self.checks))
            self._names, self._vals, self.checks[:] = [], [], []
        names, scalars, items = self._static
        return scalars[len(names):] if items else None

    def take_static(self):
        """(names, device vector, check items) of ``precompute`` / ``set_static``, removed."""
        st, self._static = self._static, None
        return st

    def device_values(self) -> Tensor:
        """float32 device vector: the put() values followed by the deferred-check flags (one launch: lazy.report)."""
        return lazy.report(self._vals, self.checks)

    def fetch(self) -> dict:
        if self._static is not None:      # values of a replayed step graph: one static device tensor, fixed names and checks
            names, scalars, items = self._static
            self._static = None
            host = scalars.tolist()
            out = dict(zip(names, host))
            checks.raise_failed(items, host[len(names):])
            return out
        if not self._vals and not self.checks:
            return {}
        host = self.device_values().tolist()
        out = dict(zip(self._names, host))
        items, self.checks[:] = list(self.checks), []
        self._names, self._vals = [], []
        checks.raise_failed(items, host[len(out):])
        return out

    def post(self, extras=()):
        """Start the asynchronous copy of this iteration's scalars (and deferred-check flags) into pinned host memory and
        return a ticket for ``wait``; ``extras`` (device tensors, e.g. the Dice counts) travel the same way and come back as
        the ticket's host tensors.  The training loop waits for iteration i's ticket only after iteration i+1 is enqueued,
        so the host never drains the GPU queue: a per-iteration ``fetch()`` left the first ~0.25 ms of every step's launches
        exposed."""
        if self._static is not None:
            names, scalars, items = self._static
            self._static = None
        elif not self._vals and not self.checks:
            names, scalars, items = [], None, []
        else:
            scalars = self.device_values()
            names, items = self._names, list(self.checks)
            self._names, self._vals, self.checks[:] = [], [], []
        payload = ([] if scalars is None else [scalars]) + list(extras)
        if not any(t.is_cuda for t in payload):
            return names, items, scalars, tuple(extras), None
        self._turn = getattr(self, "_turn", 0) ^ 1
        ring = self.__dict__.setdefault("_ring", {})
        host = []
        for i, t in enumerate(payload):
            buf = ring.get((self._turn, i))
            if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                buf = ring[(self._turn, i)] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            buf.copy_(t.detach(), non_blocking=True)
            host.append(buf)
        done = torch.cuda.Event()
        done.record()
        if scalars is None:
            return names, items, None, tuple(host), done
        return names, items, host[0], tuple(host[1:]), done

    @staticmethod
    def wait(ticket):
        """(host values, host extras) of a ``post()`` ticket; raises the deferred assertions that rode along (same errors as
        ``fetch``).  The host tensors are ring buffers: consume them before the ticket after next is posted."""
        if isinstance(ticket, stepio.Ticket):     # the iteration's one read-back block (values | flags | [overflow count], Dice counts)
            fields = ticket.io.wait(ticket)
            vals = fields["scalars"].tolist()
            out = dict(zip(ticket.names, vals))
            checks.raise_failed(ticket.items, vals[len(ticket.names):len(ticket.names) + len(ticket.items)])
            extras = [fields["inter"], fields["union"]]
            if "nonfinite" in fields:
                extras.append(fields["nonfinite"])
            elif ticket.nvals > len(ticket.names) + len(ticket.items):      # overflow count in the slot behind the flags
                extras.append(fields["scalars"][ticket.nvals - 1:ticket.nvals])
            return out, tuple(extras)
        names, items, host, extras, done = ticket
        if done is not None:
            done.synchronize()
        vals = host.tolist() if host is not None else []
        out = dict(zip(names, vals))
        checks.raise_failed(items, vals[len(names):])
        return out, extras

    def drain(self):
        """(names, device scalars, check items) recorded so far, cleared -- used while capturing a step graph."""
        names, vals, items = self._names, self._vals, list(self.checks)
        self._names, self._vals, self.checks[:] = [], [], []
        return names, vals, items

    def set_static(self, names, scalars: Tensor, items) -> None:
        self._static = (names, scalars, items)


class EvalEpocher(_num_class_mixin, _Epocher):
    """Per-patient evaluation: KL loss + 3D Dice (ref :36-73)."""

    def __init__(self, model, val_loader: T_loader, sup_criterion: T_loss, cur_epoch=0, device="cpu") -> None:
        super().__init__(model, num_batches=len(val_loader), cur_epoch=cur_epoch, device=device)
        self._val_loader = val_loader
        self._sup_criterion = sup_criterion

    def _configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters.register_meter("loss", AverageValueMeter())
        meters.register_meter("dice", UniversalDice(self.num_classes, report_axises=list(range(1, self.num_classes))))
        return meters

    @torch.no_grad()
    def _run(self, *args, **kwargs) -> Tuple[EpochResultDict, float]:
        self._model.eval()
        report_dict = EpochResultDict()
        for _, val_data in zip(self._indicator, self._val_loader):
            val_img, val_target, _file_path, _, group = self._unzip_data(val_data, self._device)
            val_logits = self._model(val_img)
            labels = val_target.squeeze(1)
            if isinstance(self._sup_criterion, KL_div) and self._sup_criterion.supports_fused():
                val_loss = self._sup_criterion.from_logits(val_logits, labels)
            else:
                val_loss = self._sup_criterion(val_logits.softmax(1), class2one_hot(labels, self.num_classes), disable_assert=True)
            _, inter, union = ops.argmax_dice(val_logits, labels, want_pred=False)
            self.meters["loss"].add(val_loss.item())
            self.meters["dice"].add_counts(inter, union, group_name=group)
            report_dict = self.meters.tracking_status()
            self._indicator.set_postfix_dict(report_dict)
        return report_dict, self.meters["dice"].summary()["DSC_mean"]

    @staticmethod
    def _unzip_data(data, device):
        return preprocess_input_with_single_transformation(data, device)


class InferenceEpocher(EvalEpocher):
    """Evaluation that also dumps image / ground truth / prediction PNGs and reports the per-class Hausdorff distance
    (ref :76-107).  The argmax + Dice counts stay fused on the device; the PNG writes and the Hausdorff distance
    (scipy) are host work, as in the reference."""

    def set_save_dir(self, save_dir):
        self._save_dir = save_dir

    def _configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters = super()._configure_meters(meters)
        meters.register_meter("hd", SurfaceMeter(C=self.num_classes, report_axises=list(range(1, self.num_classes)), metername="hausdorff"))
        return meters

    @torch.no_grad()
    def _run(self, *args, **kwargs) -> Tuple[EpochResultDict, float]:
        self._model.eval()
        report_dict = EpochResultDict()
        for _, val_data in zip(self._indicator, self._val_loader):
            val_img, val_target, file_path, _, group = self._unzip_data(val_data, self._device)
            val_logits = self._model(val_img)
            write_img_target(val_img, val_target, self._save_dir, file_path)
            write_predict(val_logits, self._save_dir, file_path)
            labels = val_target.squeeze(1)
            if isinstance(self._sup_criterion, KL_div) and self._sup_criterion.supports_fused():
                val_loss = self._sup_criterion.from_logits(val_logits, labels)
            else:
                val_loss = self._sup_criterion(val_logits.softmax(1), class2one_hot(labels, self.num_classes), disable_assert=True)
            pred, inter, union = ops.argmax_dice(val_logits, labels, want_pred=True)
            self.meters["loss"].add(val_loss.item())
            self.meters["dice"].add_counts(inter, union, group_name=group)
            try:                                   # ref: ExceptionIgnorer(RuntimeError) -- a class absent from a slice
                self.meters["hd"].add(pred, labels)
            except RuntimeError:
                pass
            report_dict = self.meters.tracking_status()
            self._indicator.set_postfix_dict(report_dict)
        return report_dict, self.meters["dice"].summary()["DSC_mean"]


class TrainEpocher(_num_class_mixin, _Epocher):
    """Supervised-only (``partial``) step (ref :110-197); the semi-supervised epochers add ``regularization``."""

    def __init__(self, model, optimizer: T_optim, labeled_loader: T_loader, unlabeled_loader: T_loader, sup_criterion: T_loss,
                 reg_weight: float, num_batches: int, cur_epoch=0, device="cpu", feature_position=None,
                 feature_importance=None) -> None:
        super().__init__(model, num_batches=num_batches, cur_epoch=cur_epoch, device=device)
        self._optimizer = optimizer
        self._labeled_loader, self._unlabeled_loader = labeled_loader, unlabeled_loader
        self._sup_criterion = sup_criterion
        self._reg_weight = reg_weight
        self._affine_transformer = TensorRandomFlip(axis=[1, 2], threshold=0.8)
        assert isinstance(feature_position, list) and isinstance(feature_position[0], str), feature_position
        assert isinstance(feature_importance, list) and isinstance(feature_importance[0], (int, float)), feature_importance
        self._feature_position, self._feature_importance = feature_position, feature_importance
        self._reducer = None  # set by miseg_amd.ddp.attach() for data-parallel runs

    def _configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters.register_meter("lr", AverageValueMeter())
        meters.register_meter("sup_loss", AverageValueMeter())
        meters.register_meter("reg_loss", AverageValueMeter())
        meters.register_meter("sup_dice", UniversalDice(self.num_classes, report_axises=list(range(1, self.num_classes))))
        return meters

    # ---- one optimisation step
    def _step(self, labeled_data, unlabeled_data):
        # host prelude: unpack the batches, draw the flip decisions (same draws, same order as the per-sample flips at ref :148-149)
        seed = random.randint(0, int(1e7))
        labeled_image, labeled_target, _, _, label_group = self._unzip_data(labeled_data, self._device)
        unlabeled_image, _unlabeled_target, *_ = self._unzip_data(unlabeled_data, self._device)
        ub = len(unlabeled_image)
        with FixRandomSeed(seed):
            decisions = self._affine_transformer.decisions(ub)
        # [flips of the UB unlabeled samples | UB zeros] (the tf branch is never re-flipped)
        flip_masks = ops.flip_masks(list(decisions) + [[False, False]] * ub)
        self._seed = seed
        io = self._io_for(labeled_image.device)
        tape = self._step_tape
        if tape is not None and (self._reducer is None or self._reducer.flat.flat_grad.is_cuda):
            ticket = tape.step(io, labeled_image, labeled_target, unlabeled_image, flip_masks)
        else:
            ticket = self._run_step(io, labeled_image, labeled_target, unlabeled_image, flip_masks)
        return ticket, None, label_group

    _io = None
    _step_tape = None
    _TAPE_DEFAULT = os.environ.get("MISEG_TAPE", "1") != "0"

    def _io_for(self, device) -> "stepio.StepIO":
        """The iteration's two host <-> device blocks (miseg_amd.stepio); with them, unless MISEG_TAPE=0, the launch tape that replays
        the iteration from one C call once it has run eagerly a few times (miseg_amd.tape.StepTape).  Both hang off the OPTIMISER,
        which outlives the per-epoch epocher objects (ref semi_seg/trainer.py:197-206 builds a new epocher every epoch): the tape
        recorded in the first epoch serves the later ones.  GPU only, like the kernels."""
        if torch.device(device).type != "cuda":
            from miseg_amd._cabi import MisegError
            raise MisegError("the train step runs on the GPU only (got CPU batches); there is no CPU fallback")
        io = self._io
        if io is None or io.device != torch.device(device):
            ctx = getattr(self._optimizer, "_miseg_step_ctx", None)
            if ctx is None or ctx["io"].device != torch.device(device):
                ctx = {"io": stepio.StepIO(device), "tape": None}
                try:
                    self._optimizer._miseg_step_ctx = ctx
                except AttributeError:
                    pass
            io = self._io = ctx["io"]
            self._step_ctx = ctx
            if ctx["tape"] is not None:
                ctx["tape"].ep = self
                self._step_tape = ctx["tape"]
            elif self._TAPE_DEFAULT and self._step_tape is None:
                self.enable_step_tape()
        return io

    def enable_step_tape(self, warmup: int = 3) -> None:
        """Replay the iteration from the library's launch tape after ``warmup`` eager iterations (one more runs eagerly while it is
        recorded).  In data-parallel runs the bucket all-reduces and the wait for them stay host calls between segments of the tape
        (miseg_amd.tape.host_call)."""
        from miseg_amd.tape import StepTape
        self.disable_step_tape()
        self._step_tape = StepTape(self, warmup=warmup)
        ctx = getattr(self, "_step_ctx", None)
        if ctx is not None:
            ctx["tape"] = self._step_tape

    def disable_step_tape(self) -> None:
        if self._step_tape is not None:
            self._step_tape.release()
        self._step_tape = None
        ctx = getattr(self, "_step_ctx", None)
        if ctx is not None:
            ctx["tape"] = None

    def _tape_signature(self):
        """What, besides shapes, decides the recorded launch list: the trainer kind and its loss coefficients (they are the constant
        gradients that seed backward)."""
        # ... and where the persistent state lives: a rebuilt flat buffer (FlatBuffers.build after the parameters moved) or moved
        # BatchNorm buffers would leave the recorded pointers dangling -- a changed address re-records instead
        opt = self._optimizer
        where = tuple((fb.flat_param.data_ptr(), fb.flat_grad.data_ptr()) for fb in getattr(opt, "_flats", []) if fb.flat_param is not None)
        where += tuple(t.data_ptr() for t in getattr(opt, "_m", []) if t is not None)
        buf = next(iter(self._model.buffers()), None)
        return (type(self).__name__, float(self._reg_weight), getattr(self, "_cons_weight", None), getattr(self, "_iic_weight", None),
                tuple(self._feature_importance), tuple(self._feature_position), type(self._sup_criterion).__name__,
                type(getattr(self, "_reg_criterion", None)).__name__, where, None if buf is None else buf.data_ptr(), id(self._model))

    def _loss_scale(self) -> float:
        """Current loss scale: 1 unless the activations are IEEE half (BASELINE configs[4]); dynamic from its initial value."""
        scale = unet_ops.loss_scale_of(self._model)
        if scale == 1.0:
            return 1.0
        if not hasattr(self._optimizer, "grad_scale"):
            raise RuntimeError("Arch.compute_dtype=float16 needs the fused Adam (Optim.name=Adam), which undoes the loss scale")
        if self._optimizer.loss_scaler is None:
            from miseg_amd.flat import LossScaler
            self._optimizer.loss_scaler = LossScaler(scale)
        return self._optimizer.loss_scaler.scale

    def _stage(self, io, flip_masks) -> int:
        """Host half of an iteration: the optimiser's step counter / scalars, the loss scale and the flip masks into the next pinned
        slot of the step block.  No device work (a replayed launch tape calls just this)."""
        rows = self._optimizer.host_step() if hasattr(self._optimizer, "host_step") else []
        scale = self._loss_scale()
        if hasattr(self._optimizer, "grad_scale"):
            self._optimizer.grad_scale = scale
        return io.stage(flip_masks, rows, scale)

    def _run_step(self, io, labeled_image: Tensor, labeled_target: Tensor, unlabeled_image: Tensor, flip_masks):
        """One eager iteration on the GPU: upload, forward, losses, backward, optimiser, read-back.  Returns the read-back ticket."""
        io.upload(self._stage(io, flip_masks))
        stepio.CURRENT = io
        try:
            self._device_step(labeled_image, labeled_target, unlabeled_image, io.flips())
            return self._post(io)
        finally:
            stepio.CURRENT = None

    def _after_replay(self) -> None:
        unet_ops.PACK_CACHE.invalidate()       # the replay ran Adam: eager users of the weights (evaluation) must re-pack

    def _post(self, io):
        """The iteration's report (if the guarded optimiser launch has not computed it already) and the one device -> host copy."""
        self._pending.precompute()
        st = self._pending.take_static()
        names, scalars, items = st if st is not None else ([], None, [])
        rep = getattr(io, "last_report", None)
        if scalars is not None and (rep is None or scalars.data_ptr() != rep.data_ptr()):
            # evaluated outside the read-back block (a value that does not live in the scalar arena): copy it in
            rep = io.out("scalars", (scalars.numel() + 1,), torch.float32)
            rep[:scalars.numel()].copy_(scalars)
        nvals = 0 if rep is None else rep.numel() - (0 if getattr(self._optimizer, "last_nonfinite", None) is not None and
                                                     self._optimizer.last_nonfinite.data_ptr() == rep[-1:].data_ptr() else 1)
        return io.post(names, items, nvals)

    def _before_forward(self, ub: int) -> None:   # hooks for epochers that start work while the network is still running
        pass

    def _after_forward(self) -> None:
        pass

    def _device_step(self, labeled_image: Tensor, labeled_target: Tensor, unlabeled_image: Tensor, flips2: Tensor, seed: int = None):
        """Everything of the iteration that runs on the GPU between the upload and the read-back: forward, losses, backward,
        optimiser kernel, Dice counts.  No host synchronisation, no host->device traffic, and -- for the shipped trainers -- no launch
        outside the library (so the iteration can be replayed from the launch tape); the host half (``_stage``) has run."""
        seed = self._seed if seed is None else seed
        lb, ub = len(labeled_image), len(unlabeled_image)
        self._flips2 = flips2
        flips = flips2[:ub]
        if labeled_image.dtype == unlabeled_image.dtype == torch.float32 and labeled_image.dim() == 4 and \
                labeled_image.shape[1:] == unlabeled_image.shape[1:] and labeled_image.is_cuda:
            # [labeled | unlabeled | flip(unlabeled)] in one launch (ref :148-153: per-sample flips, stack, cat)
            net = getattr(self._model, "module", self._model)
            batch = ops.cat_flip(labeled_image, unlabeled_image, flips,
                                 stem_dtype=getattr(net, "compute_dtype", None) if getattr(net, "input_dim", None) == 1 else None)
            unlabeled_image_tf = batch[lb + ub:]
        else:
            unlabeled_image_tf = ops.flip(unlabeled_image, flips)
            batch = torch.cat([labeled_image, unlabeled_image, unlabeled_image_tf], dim=0)
        assert unlabeled_image_tf.shape == unlabeled_image.shape

        with checks.deferred(self._pending.checks):   # simplex / NaN assertions are raised at this iteration's fetch()
            self._before_forward(ub)
            try:
                predict_logits = self._model(batch)
            finally:
                self._after_forward()
        label_logits, unlabel_logits, unlabel_tf_logits = ops.split_rows(predict_logits, [lb, ub, ub])
        labels = labeled_target.squeeze(1)
        with checks.deferred(self._pending.checks):
            if isinstance(self._sup_criterion, KL_div) and self._sup_criterion.supports_fused():
                sup_loss = self._sup_criterion.from_logits(label_logits, labels)
            else:
                sup_loss = self._sup_criterion(label_logits.softmax(1), class2one_hot(labels, self.num_classes))
            reg_fn = self.regularization
            fused = getattr(reg_fn, "_miseg_fused", False)
            reg_loss = reg_fn(
                unlabeled_tf_logits=unlabel_tf_logits,
                unlabeled_logits_tf=None if fused else ops.flip(unlabel_logits, flips),
                seed=seed, unlabeled_image=unlabeled_image, unlabeled_image_tf=unlabeled_image_tf,
                unlabeled_logits=unlabel_logits, flips=flips, num_unlabeled=ub,
            )
        total_loss = sup_loss + self._reg_weight * reg_loss
        # Everything the read-back needs from the FORWARD pass is launched here, before backward: the Dice counts and the iteration's
        # report (meter values + the simplex / NaN flags that guard the update).  Issued after backward they sat behind the optimiser's
        # wait for the weight-gradient stream, i.e. on the step's tail (~19 us); here they run while the main stream waits for the IIC chain.
        def forward_readback():
            with torch.no_grad():
                self._pending.put("sup_loss", sup_loss)
                self._pending.put("reg_loss", reg_loss)
                dice = ops.argmax_dice(label_logits.detach(), labels, want_pred=False, to_host=True)
            return dice[1], dice[2], (self._pending.precompute() if _GUARD_STEP and hasattr(self._optimizer, "apply") else None)
        if _READBACK_EARLY:
            inter, union, guard = forward_readback()
        self._optimizer.zero_grad()
        if self._reducer is not None:
            self._reducer.prepare()
        # fp16 storage mode: backward is seeded with the loss scale; the fused Adam reads the gradients as grad / scale, counts their
        # non-finite entries and skips the update if there are any; the scale follows (flat.LossScaler, one iteration late)
        scale = self._loss_scale()
        if isinstance(total_loss, LinearLoss):
            total_loss.backward(scale)
        elif scale != 1.0:
            (total_loss * scale).backward()
        else:
            total_loss.backward()
        if self._reducer is not None:
            self._reducer.finish()
        io = stepio.CURRENT
        if not _READBACK_EARLY:
            inter, union, guard = forward_readback()
        if hasattr(self._optimizer, "apply"):
            # The report's flags guard the update on the device: the host raises a failed check one iteration late, but it has not moved
            # the weights (the reference raises before backward).
            self._optimizer.apply(guard=guard, io=io)
        else:
            self._optimizer.step()
        self._overflow = getattr(self._optimizer, "last_nonfinite", None)     # device float[1] in the fp16 mode, else None
        return inter, union

    def _run(self, *args, **kwargs) -> EpochResultDict:
        self.meters["lr"].add(get_lrs_from_optimizer(self._optimizer)[0])
        self._model.train()
        assert self._model.training, self._model.training
        report_dict = {}
        self._pending = _Pending()
        with FeatureExtractor(self._model, self._feature_position) as self._fextractor:
            for _, labeled_data, unlabeled_data in zip(self._indicator, self._labeled_loader, self._unlabeled_loader):
                self._after_step(*self._step(labeled_data, unlabeled_data))
                report_dict = self.meters.tracking_status()
                self._indicator.set_postfix_dict(report_dict)
            self._flush_records()
            report_dict = self.meters.tracking_status()
        return report_dict

    # The iteration's single host read-back (ref: the .item() calls of semi_seg/epocher.py:115-121) is taken one iteration
    # late: iteration i's scalars travel to pinned memory asynchronously and are recorded after iteration i+1 is enqueued, so
    # the GPU always has the next step queued.  Every iteration is still recorded (the last one by _flush_records), a NaN loss
    # or a failed deferred assertion raises one iteration later.  MISEG_DEFER_FETCH=0 restores the synchronous read-back.
    _DEFER_FETCH = os.environ.get("MISEG_DEFER_FETCH", "1") != "0"
    _inflight = None

    _overflow = None

    def _after_step(self, inter, union: Optional[Tensor], label_group) -> None:
        if isinstance(inter, stepio.Ticket):      # the GPU path: the read-back is already under way (_post)
            ticket = inter
        else:
            extras = (inter, union) if self._overflow is None else (inter, union, self._overflow)
            ticket = self._pending.post(extras)
        prev, self._inflight = self._inflight, (ticket, label_group)
        if not self._DEFER_FETCH:
            self._flush_records()
        elif prev is not None:
            self._record_ticket(*prev)

    def _flush_records(self) -> None:
        last, self._inflight = self._inflight, None
        if last is not None:
            self._record_ticket(*last)

    def _record_ticket(self, ticket, label_group) -> None:
        host, extras = self._pending.wait(ticket)
        inter, union = extras[0], extras[1]
        if len(extras) > 2:         # fp16 mode: the gradient's non-finite count -> the loss scale of the iterations still to be enqueued
            scaler = self._optimizer.loss_scaler
            before, bad = scaler.scale, float(extras[2][0])
            scaler.update(bad, getattr(ticket, "scale", None))
            if (bad != 0.0 or bad != bad) and hasattr(self._optimizer, "step_skipped"):
                self._optimizer.step_skipped()          # the device skipped that update: it does not count towards the bias corrections
            if scaler.scale < before:
                import warnings
                warnings.warn(f"fp16 gradient overflow: that optimiser step was skipped, loss scale {before:g} -> {scaler.scale:g}")
        self._record(host, inter.clone(), union.clone(), label_group)   # the meters keep them; the pinned ring is reused

    def _record(self, host: dict, inter: Tensor, union: Tensor, label_group) -> None:
        self.meters["sup_loss"].add(host["sup_loss"])
        self.meters["sup_dice"].add_counts(inter, union, group_name=label_group)
        self.meters["reg_loss"].add(host["reg_loss"])

    @staticmethod
    def _unzip_data(data, device):
        (image, target), _, filename, partition, group = preprocess_input_with_twice_transformation(data, device)
        return image, target, filename, partition, group

    @_fused
    def regularization(self, *args, **kwargs):
        return LinearLoss([])      # symbolic zero (ref :194-197 returns a zero tensor): no kernel, reported as 0


class UDATrainEpocher(TrainEpocher):
    """Consistency between f(flip(x)) and flip(f(x)).detach() (ref :200-226)."""

    def __init__(self, model, optimizer, labeled_loader, unlabeled_loader, sup_criterion, reg_criterion: T_loss, reg_weight: float,
                 num_batches: int, cur_epoch: int = 0, device="cpu", feature_position=None, feature_importance=None) -> None:
        super().__init__(model, optimizer, labeled_loader, unlabeled_loader, sup_criterion, reg_weight, num_batches, cur_epoch,
                         device, feature_position, feature_importance)
        self._reg_criterion = reg_criterion

    def _configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters = super()._configure_meters(meters)
        meters.register_meter("uda", AverageValueMeter())
        return meters

    def _uda(self, unlabeled_tf_logits: Tensor, unlabeled_logits: Tensor, flips: Tensor) -> Tensor:
        if isinstance(self._reg_criterion, nn.MSELoss):
            loss = LinearLoss.of(ops.softmax_mse(unlabeled_tf_logits, unlabeled_logits, flips))  # flip + 2 softmaxes + MSE fused
        elif isinstance(self._reg_criterion, KL_div) and self._reg_criterion.supports_fused():   # `UDARegCriterion.name: kl`
            loss = LinearLoss.of(ops.softmax_kl_consistency(unlabeled_tf_logits, unlabeled_logits, flips))
        else:  # any other criterion object: called as the reference calls it, on materialised operands
            loss = self._reg_criterion(unlabeled_tf_logits.softmax(1), ops.flip(unlabeled_logits, flips).softmax(1).detach())
        self._pending.put("uda", loss)
        return loss

    @_fused
    def regularization(self, unlabeled_tf_logits: Tensor, unlabeled_logits_tf: Tensor = None, seed=None, *args,
                       unlabeled_logits: Tensor = None, flips: Tensor = None, **kwargs):
        return self._uda(unlabeled_tf_logits, unlabeled_logits, flips)

    def _record(self, host, inter, union, label_group):
        super()._record(host, inter, union, label_group)
        if "uda" in host:
            self.meters["uda"].add(host["uda"])


class IICTrainEpocher(TrainEpocher):
    """Global (encoder taps) + local (decoder taps) IIC mutual information (ref :229-284)."""

    def _configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters = super()._configure_meters(meters)
        meters.register_meter("mi", AverageValueMeter())
        meters.register_meter("individual_mis", MultipleAverageValueMeter())
        return meters

    def __init__(self, model, projectors_wrapper: ProjectorWrapper, optimizer, labeled_loader, unlabeled_loader, sup_criterion,
                 IIDSegCriterionWrapper: IICLossWrapper, reg_weight: float, num_batches: int, cur_epoch: int = 0, device="cpu",
                 feature_position=None, feature_importance=None) -> None:
        super().__init__(model, optimizer, labeled_loader, unlabeled_loader, sup_criterion, reg_weight, num_batches, cur_epoch,
                         device, feature_position, feature_importance)
        assert projectors_wrapper.feature_names == self._feature_position
        self._projectors_wrapper = projectors_wrapper
        assert IIDSegCriterionWrapper.feature_names == self._feature_position
        self._IIDSegCriterionWrapper = IIDSegCriterionWrapper

    # ---- The IIC branch runs on its own HIP stream.  (1) Each tapped feature is turned into its MI loss the moment the
    # forward hook delivers it (FeatureExtractor.on_feature), so heads + joint of the Up_conv3 tap overlap the rest of the
    # decoder on the main stream; (2) autograd replays backward nodes on the stream of their forward op, so the
    # matrix-core-bound local-MI backward of the earlier taps overlaps the HBM-bound BatchNorm / weight-gradient kernels of
    # the later decoder blocks.  Backward runs the most recently created chain first, i.e. the last tap -- the one the main
    # stream has to wait for -- goes first.
    def _use_side_stream(self, dev) -> bool:
        if dev.type != "cuda" or os.environ.get("MISEG_IIC_STREAM", "1") != "1":
            return False
        # under hipGraph capture the fork/join (wait_stream both ways) is captured as graph dependencies
        return os.environ.get("MISEG_GRAPH_STREAMS", "1") == "1" or not torch.cuda.is_current_stream_capturing()

    def _side(self, dev):
        side = getattr(self, "_iic_stream", None)
        if side is None:
            side = self._iic_stream = torch.cuda.Stream(device=dev)
        red = getattr(self, "_reducer", None)
        if red is not None and side not in red.producer_streams:
            red.producer_streams.append(side)
            red.producer_streams.append(torch.cuda.current_stream(dev))
        return side

    def _before_forward(self, ub: int) -> None:
        self._early, self._ub_now = {}, ub
        fx = getattr(self, "_fextractor", None)
        if fx is not None and hasattr(fx, "on_feature") and self._use_side_stream(self._flips2.device):
            fx.on_feature = self._on_feature

    def _after_forward(self) -> None:
        fx = getattr(self, "_fextractor", None)
        if fx is not None and hasattr(fx, "on_feature"):
            fx.on_feature = None

    def _on_feature(self, name: str, feature: Tensor) -> None:
        idx = self._feature_position.index(name)
        projector, criterion = list(self._projectors_wrapper)[idx], list(self._IIDSegCriterionWrapper)[idx]
        main, side = torch.cuda.current_stream(feature.device), self._side(feature.device)
        ops.wait_stream(side, main)
        _keep(feature, side)
        with torch.cuda.stream(side):
            self._early[name] = self._tap_loss(feature, projector, criterion, self._flips2, self._ub_now)

    def _iic(self, flips: Tensor, ub: int):
        if not self._use_side_stream(flips.device):
            return self._iic_body(flips, ub)
        main, side = torch.cuda.current_stream(flips.device), self._side(flips.device)
        ops.wait_stream(side, main)
        with torch.cuda.stream(side):
            out = self._iic_body(flips, ub)
        ops.wait_stream(main, side)
        for t, _ in LinearLoss.of(out).terms:      # produced on `side`, read on `main`: keep the allocator informed
            _keep(t, main)
        return out

    def _tap_loss(self, feature: Tensor, projector, criterion, flips2: Tensor, ub: int):
        total = feature.shape[0]
        # last 2*UB samples of the tap: [features(unlabeled) | features(flip(unlabeled))]   (ref :258-259)
        src = ops.arange_i32(total - 2 * ub, total, feature.device)
        if isinstance(projector, ClusterHead):  # encoder tap: global pooling is flip-invariant (ref :261-262)
            probs = projector.forward_gathered(feature, src)                       # [S, 2UB, K]
            if _DEBUG_ASSERTS:
                from contrastyou.losses.iic_loss import simplex
                assert simplex(probs.flatten(0, 1))
            if probs.shape[1] != 2 * ub:
                raise RuntimeError(f"encoder tap: {probs.shape[1]} rows per sub-head, expected 2 x {ub}")
            per_head, _, _ = ops.global_mi_pair(probs, criterion.lamb)
            return LinearLoss.mean(per_head)
        # decoder tap: replay the flip on features(unlabeled) (ref :264-266), fused into the head
        probs = projector.forward_gathered(feature, src, flips2)  # [S, 2UB, K, H, W]
        if hasattr(criterion, "forward_heads"):   # all sub-heads as one autograd node (gradient lands in one buffer)
            return criterion.forward_heads(probs, ub, lazy=True)
        return average_iter([criterion(p[:ub], p[ub:]) for p in probs])

    def _iic_body(self, flips: Tensor, ub: int):
        flips2 = self._flips2 if getattr(self, "_flips2", None) is not None and len(self._flips2) == 2 * ub \
            else torch.cat([flips, torch.zeros_like(flips)])
        early = getattr(self, "_early", {})
        losses = []
        for name, feature, projector, criterion in zip(self._feature_position, self._fextractor, self._projectors_wrapper,
                                                       self._IIDSegCriterionWrapper):
            loss = early.pop(name, None)      # already launched from the forward hook (side stream)
            losses.append(loss if loss is not None else self._tap_loss(feature, projector, criterion, flips2, ub))
        reg_loss = weighted_average_iter(losses, self._feature_importance)
        self._pending.put("mi", -reg_loss)
        for name, v in zip(self._feature_position, losses):
            self._pending.put("mi/" + name, -v)
        return reg_loss

    @_fused
    def regularization(self, unlabeled_tf_logits: Tensor, unlabeled_logits_tf: Tensor = None, seed: int = None, *args,
                       flips: Tensor = None, num_unlabeled: int = None, **kwargs):
        return self._iic(flips, num_unlabeled if num_unlabeled is not None else len(unlabeled_tf_logits))

    def _record(self, host, inter, union, label_group):
        super()._record(host, inter, union, label_group)
        if "mi" in host:
            self.meters["mi"].add(host["mi"])
            self.meters["individual_mis"].add(**{n: host["mi/" + n] for n in self._feature_position})


class UDAIICEpocher(IICTrainEpocher):
    """``cons_weight * UDA + iic_weight * IIC`` with reg_weight forced to 1 (ref :287-323)."""

    def __init__(self, model, projectors_wrapper: ProjectorWrapper, optimizer, labeled_loader, unlabeled_loader, sup_criterion,
                 reg_criterion: T_loss, IIDSegCriterion: T_loss, num_batches: int, cur_epoch: int = 0, device="cpu",
                 feature_position=None, feature_importance=None, cons_weight=1, iic_weight=0.1) -> None:
        super().__init__(model, projectors_wrapper, optimizer, labeled_loader, unlabeled_loader, sup_criterion, IIDSegCriterion,
                         1.0, num_batches, cur_epoch, device, feature_position, feature_importance)
        self._cons_weight, self._iic_weight = cons_weight, iic_weight
        self._reg_criterion = reg_criterion

    def _configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters = super()._configure_meters(meters)
        for name in ("uda", "iic_weight", "uda_weight"):
            meters.register_meter(name, AverageValueMeter())
        return meters

    _uda = UDATrainEpocher._uda

    @_fused
    def regularization(self, unlabeled_tf_logits: Tensor, unlabeled_logits_tf: Tensor = None, seed: int = None, *args,
                       unlabeled_logits: Tensor = None, flips: Tensor = None, num_unlabeled: int = None, **kwargs):
        iic_loss = self._iic(flips, num_unlabeled if num_unlabeled is not None else len(unlabeled_tf_logits))
        cons_loss = self._uda(unlabeled_tf_logits, unlabeled_logits, flips)
        return self._cons_weight * cons_loss + self._iic_weight * iic_loss

    def _record(self, host, inter, union, label_group):
        super()._record(host, inter, union, label_group)
        self.meters["uda"].add(host["uda"])
        # host-side bookkeeping lives here, not in regularization(): that one is part of the captured device step
        self.meters["iic_weight"].add(self._iic_weight)
        self.meters["uda_weight"].add(self._cons_weight)
