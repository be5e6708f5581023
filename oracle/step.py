"""Oracle: one full ``partial`` / ``udaiic`` train step, restated as one function.

Follows ``semi_seg/epocher.py``: TrainEpocher._run :137-188 (one forward on
cat[labeled, unlabeled, flip(unlabeled)], supervised KL, regulariser, backward, Adam),
UDATrainEpocher.regularization :215-226, IICTrainEpocher.regularization :249-284,
UDAIICEpocher.regularization :308-323, and ``contrastyou/helper/utils.py:46-56``
for the (weighted) averages.  Test infrastructure only.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
from torch import Tensor

from . import heads as H
from . import iic
from . import losses as L
from . import unet as U

ENC = set(U.ENCODER)


def weighted_average(values, weights):
    """weighted_average_iter (helper/utils.py:54-56): note the +1e-16 in the denominator."""
    return sum(v * w for v, w in zip(values, weights)) / (sum(weights) + 1e-16)


class StepState:
    """Everything the step mutates: U-Net state_dict, per-feature head weights, Adam moments."""

    def __init__(self, model_sd, head_sds: "OrderedDict[str, OrderedDict[str, Tensor]]", lr=1e-7, weight_decay=1e-5):
        self.model = model_sd
        self.heads = head_sds
        self.lr, self.wd = lr, weight_decay
        self.t = 0
        self._params = [(k, model_sd) for k in U.trainable_keys(model_sd)]
        for f, sd in head_sds.items():
            self._params += [(k, sd) for k in sd]
        self.m = [torch.zeros_like(sd[k]) for k, sd in self._params]
        self.v = [torch.zeros_like(sd[k]) for k, sd in self._params]

    def params(self):
        return [sd[k] for k, sd in self._params]

    def names(self):
        out = []
        for k, sd in self._params:
            if sd is self.model:
                out.append(k)
            else:
                f = next(n for n, h in self.heads.items() if h is sd)
                out.append(f"{f}/{k}")
        return out


def train_step(state: StepState, labeled_img: Tensor, labeled_tgt: Tensor, unlabeled_img: Tensor, seed: int,
               mode: str = "udaiic", feature_names=("Conv5", "Up_conv3", "Up_conv2"),
               feature_importance=(0.5, 0.25, 0.25), paddings=(1, 3), patch_sizes=(1024, 1024),
               cons_weight: float = 5.0, iic_weight: float = 0.1, num_classes: int = 4, do_update: bool = True, unet_fn=None,
               head_normalize: bool = False, uda_criterion: str = "mse"):
    """Returns a dict of scalars (meter values) and, if requested, gradients by parameter name.  ``unet_fn`` swaps the network
    evaluation (default ``oracle.unet.unet_forward``; ``unet_forward_bf16_autograd`` emulates the bf16 kernels' rounding points)."""
    params = state.params()
    for p in params:
        p.requires_grad_(True)
        p.grad = None
    lb, ub = labeled_img.shape[0], unlabeled_img.shape[0]
    decisions = L.flip_decisions(seed, ub)
    unlabeled_tf = L.apply_flips(unlabeled_img, decisions)                       # epocher.py:148-149
    logits, feats = (unet_fn or U.unet_forward)(state.model, torch.cat([labeled_img, unlabeled_img, unlabeled_tf], 0), True)
    label_logits, unlabel_logits, unlabel_tf_logits = torch.split(logits, [lb, ub, ub], 0)
    unlabel_logits_tf = L.apply_flips(unlabel_logits, decisions)                 # epocher.py:160-161
    onehot = L.class2one_hot(labeled_tgt.squeeze(1), num_classes)
    sup = L.kl_div(label_logits.softmax(1), onehot)                              # epocher.py:165-166
    out = {"sup_loss": sup}
    reg = torch.zeros((), dtype=sup.dtype)
    if mode in ("iic", "udaiic"):
        per_feature = []
        dec_i = 0
        for fname in feature_names:
            feat = feats[fname]
            u = feat[feat.shape[0] - 2 * ub:]
            f_u, f_tf = torch.chunk(u, 2, 0)                                      # epocher.py:258-259
            hsd = state.heads[fname]
            if fname in ENC:                                                      # epocher.py:261-262
                probs = H.cluster_head(hsd, torch.cat([f_u, f_tf], 0), normalize=head_normalize)
                pairs = [torch.chunk(p, 2, 0) for p in probs]
                ls = [iic.iid_loss(a, b)[0] for a, b in pairs]                    # semi_seg/_utils.py:12-15
            else:
                f_u_tf = L.apply_flips(f_u, decisions)                            # epocher.py:264-266
                probs = H.local_cluster_head(hsd, torch.cat([f_u_tf, f_tf], 0), normalize=head_normalize)
                pairs = [torch.chunk(p, 2, 0) for p in probs]
                ls = [iic.iid_seg_small_patch_loss(a, b, paddings[dec_i], patch_sizes[dec_i]) for a, b in pairs]
                dec_i += 1
            per_feature.append(sum(ls) / float(len(ls)))
        iic_loss = weighted_average(per_feature, list(feature_importance))
        out["mi"] = -iic_loss
        for fname, v in zip(feature_names, per_feature):
            out[f"mi/{fname}"] = -v
    if mode in ("uda", "udaiic"):
        if uda_criterion == "kl":                                                 # trainer.py:137,194: KL_div()(softmax(a), softmax(b).detach())
            uda = L.kl_div(unlabel_tf_logits.softmax(1), unlabel_logits_tf.softmax(1).detach())
        else:
            uda = L.softmax_mse(unlabel_tf_logits, unlabel_logits_tf)             # epocher.py:221-224
        out["uda"] = uda
    if mode == "udaiic":
        reg = cons_weight * uda + iic_weight * iic_loss                           # epocher.py:323
    elif mode == "uda":
        reg = uda
    elif mode == "iic":
        reg = iic_loss
    out["reg_loss"] = reg
    weight = {"partial": 0.0, "uda": cons_weight, "iic": iic_weight, "udaiic": 1.0}[mode]
    total = sup + weight * reg                                                    # epocher.py:175
    out["total"] = total
    total.backward()
    grads = OrderedDict((n, (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)))
                        for n, p in zip(state.names(), params))
    with torch.no_grad():
        pred = label_logits.max(1)[1]
        out["pred"] = pred
        if do_update:
            state.t += 1
            L.adam_step([p.data for p in params], [grads[n] for n in state.names()], state.m, state.v, state.t,
                        state.lr, weight_decay=state.wd)
    for p in params:
        p.requires_grad_(False)
    scalars = {k: (float(v.detach()) if torch.is_tensor(v) and v.dim() == 0 else v) for k, v in out.items()}
    return scalars, grads
