#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference; the GPU box never has it):

    python tests/golden/make_golden.py

What it does: copies the reference's python packages and unzips its vendored wheel into a
throw-away temp dir (the reference tree is read-only and creates ``.data`` dirs on import),
installs a small compatibility shim for the torch-1.6 / py3.7-era imports (removed aliases
such as ``torch._six`` and ``collections.MutableMapping``; empty stand-ins for packages that do no
arithmetic on this path: termcolor, torchvision, tensorboardX, ...), imports the reference,
feeds it seeded inputs and stores inputs + outputs as small ``.npz`` fixtures.  Only DATA is
written to the repo; no reference source is copied.  See SURVEY.md section 8(c) / Appendix A.
"""
from __future__ import annotations

import collections
import collections.abc
import importlib.abc
import importlib.machinery
import os
import random
import shutil
import sys
import tempfile
import types
import zipfile

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)                                   # synth.py
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))  # repo root (oracle/ supplies synthetic state dicts)
import synth  # noqa: E402


# ----------------------------------------------------------------------------- shim + import
def _install_shim():
    for n in ("MutableMapping", "Mapping", "Iterator", "Iterable"):
        if not hasattr(collections, n):
            setattr(collections, n, getattr(collections.abc, n))
    six = types.ModuleType("torch._six")
    six.container_abcs, six.int_classes, six.string_classes, six.inf = collections.abc, int, str, float("inf")
    sys.modules["torch._six"] = six
    torch._six = six

    class _Anything:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

        def __getattr__(self, k):
            return _Anything()

    class _Hollow(types.ModuleType):
        __all__: list = []

        def __getattr__(self, k):
            if k.startswith("__") and k.endswith("__"):
                raise AttributeError(k)
            return _Anything

    roots = {"termcolor", "torch_optimizer", "torchvision", "tensorboardX", "skimage", "medpy", "cv2", "easydict",
             "gdown", "SimpleITK"}

    class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path, target=None):
            if name.split(".")[0] in roots:
                return importlib.machinery.ModuleSpec(name, self, is_package=True)

        def create_module(self, spec):
            m = _Hollow(spec.name)
            m.__path__ = []
            return m

        def exec_module(self, m):
            pass

    sys.meta_path.append(_Finder())
    import termcolor
    termcolor.colored = lambda s, *a, **k: s
    import tqdm.utils as tu
    if not hasattr(tu, "_OrderedDict"):
        tu._OrderedDict = collections.OrderedDict


def import_reference():
    scratch = tempfile.mkdtemp(prefix="miseg_ref_")
    for pkg in ("contrastyou", "semi_seg", "config"):
        shutil.copytree(os.path.join(REF, pkg), os.path.join(scratch, pkg))
    with zipfile.ZipFile(os.path.join(REF, "deepclustering2-2.0.0-py3-none-any.whl")) as z:
        z.extractall(scratch)
    _install_shim()
    sys.path.insert(0, scratch)
    return scratch


def np_(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np_(v) for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def sd_arrays(prefix, sd):
    return {f"{prefix}{k}": v for k, v in sd.items()}


# ----------------------------------------------------------------------------- generators
# Inputs come from tests/golden/synth.py (bit-reproducible); big outputs are stored as fingerprints.
T = torch.from_numpy


def put_fp(out, key, tensor):
    out.update(synth.fp_pack(key, synth.fingerprint(np_(tensor), key)))


def gen_iic():
    from contrastyou.losses.iic_loss import IIDLoss, IIDSegmentationLoss, IIDSegmentationSmallPathLoss, \
        patch_generator
    import torch.nn.functional as F
    out = {}
    # 1. global MI: (loss, loss_no_lamb, P) and input grads
    for npdt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        for (n, k) in ((7, 5), (16, 20)):
            key = f"global_{tag}_n{n}_k{k}"
            x = T(synth.probs(key + "/x", (n, k), npdt)).requires_grad_(True)
            y = T(synth.probs(key + "/y", (n, k), npdt)).requires_grad_(True)
            loss, loss_nl, p = IIDLoss()(x, y)
            gx, gy = torch.autograd.grad(loss, [x, y])
            out.update({f"{key}/loss": loss, f"{key}/loss_no_lamb": loss_nl, f"{key}/joint": p,
                        f"{key}/gx": gx, f"{key}/gy": gy})
    # 2. local MI: loss, raw conv joint [K,K,T,T], input-grad fingerprints
    for npdt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        for (n, k, h, w, p) in ((3, 5, 12, 10, 2), (4, 20, 32, 32, 1), (4, 20, 32, 32, 3), (2, 8, 64, 64, 3)):
            key = f"local_{tag}_n{n}_k{k}_h{h}_w{w}_p{p}"
            x = T(synth.probs(key + "/x", (n, k, h, w), npdt)).requires_grad_(True)
            y = T(synth.probs(key + "/y", (n, k, h, w), npdt)).requires_grad_(True)
            loss = IIDSegmentationLoss(padding=p)(x, y)
            gx, gy = torch.autograd.grad(loss, [x, y])
            raw = F.conv2d(x.detach().permute(1, 0, 2, 3).contiguous(),
                           weight=y.detach().permute(1, 0, 2, 3).contiguous(), padding=(p, p))
            out.update({f"{key}/loss": loss, f"{key}/raw_kktt": raw})
            put_fp(out, f"{key}/gx", gx)
            put_fp(out, f"{key}/gy", gy)
    # 3. patch-averaged local MI (incl. the reference's own __main__ geometry: 100x100 map, patch 32)
    for npdt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        for (n, k, h, w, p, patch, use_mask) in ((2, 4, 100, 100, 1, 32, False), (2, 4, 100, 100, 1, 32, True),
                                                   (2, 6, 64, 64, 2, 1024, False), (1, 3, 512, 512, 3, 128, False),
                                                   (2, 5, 48, 40, 1, 16, True)):
            key = f"patch_{tag}_n{n}_k{k}_h{h}_w{w}_p{p}_ps{patch}_m{int(use_mask)}"
            x = T(synth.probs(key + "/x", (n, k, h, w), npdt)).requires_grad_(True)
            y = T(synth.probs(key + "/y", (n, k, h, w), npdt)).requires_grad_(True)
            m = T(synth.mask(key + "/mask", (n, 1, h, w)).astype(npdt)) if use_mask else None
            loss = IIDSegmentationSmallPathLoss(padding=p, patch_size=patch)(x, y, m)
            gx, gy = torch.autograd.grad(loss, [x, y])
            out[f"{key}/loss"] = loss
            put_fp(out, f"{key}/gx", gx)
            put_fp(out, f"{key}/gy", gy)
    # 5. the same three losses on PEAKED, correlated inputs (synth.peaked_pair): MI of O(0.1 .. 1), where "1e-5 relative" can be
    #    demanded of the loss literally (the near-uniform cases above have losses of 1e-3 .. 1e-6 = differences of O(1) entropies)
    for npdt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        for (n, k) in ((16, 20), (64, 5)):
            key = f"gpeak_{tag}_n{n}_k{k}"
            xs, ys = synth.peaked_pair(key, (n, k), npdt)
            x, y = T(xs).requires_grad_(True), T(ys).requires_grad_(True)
            loss, loss_nl, p = IIDLoss()(x, y)
            gx, gy = torch.autograd.grad(loss, [x, y])
            out.update({f"{key}/loss": loss, f"{key}/loss_no_lamb": loss_nl, f"{key}/joint": p, f"{key}/gx": gx, f"{key}/gy": gy})
        for (n, k, h, w, p) in ((3, 5, 12, 10, 2), (4, 20, 32, 32, 1), (4, 20, 32, 32, 3), (2, 8, 64, 64, 3)):
            key = f"lpeak_{tag}_n{n}_k{k}_h{h}_w{w}_p{p}"
            xs, ys = synth.peaked_pair(key, (n, k, h, w), npdt)
            x, y = T(xs).requires_grad_(True), T(ys).requires_grad_(True)
            loss = IIDSegmentationLoss(padding=p)(x, y)
            gx, gy = torch.autograd.grad(loss, [x, y])
            out[f"{key}/loss"] = loss
            put_fp(out, f"{key}/gx", gx)
            put_fp(out, f"{key}/gy", gy)
        for (n, k, h, w, p, patch) in ((2, 4, 100, 100, 1, 32), (1, 20, 96, 96, 3, 32)):
            key = f"ppeak_{tag}_n{n}_k{k}_h{h}_w{w}_p{p}_ps{patch}"
            xs, ys = synth.peaked_pair(key, (n, k, h, w), npdt)
            x, y = T(xs).requires_grad_(True), T(ys).requires_grad_(True)
            loss = IIDSegmentationSmallPathLoss(padding=p, patch_size=patch)(x, y, None)
            gx, gy = torch.autograd.grad(loss, [x, y])
            out[f"{key}/loss"] = loss
            put_fp(out, f"{key}/gx", gx)
            put_fp(out, f"{key}/gy", gy)
    # 4. patch_generator geometry
    for (h, patch) in ((100, 32), (64, 1024), (512, 128), (224, 1024), (48, 16), (33, 16)):
        fm = torch.arange(h * h, dtype=torch.float32).view(1, 1, h, h)
        wins = []
        for pt in patch_generator(fm, (patch, patch), (patch // 2, patch // 2)):
            v0 = int(pt[0, 0, 0, 0])
            wins.append([v0 // h, v0 // h + pt.shape[2], v0 % h, v0 % h + pt.shape[3]])
        out[f"patchgeom_h{h}_ps{patch}/windows"] = np.asarray(wins, dtype=np.int64)
    save("iic", **out)


def gen_heads():
    from contrastyou.trainer._utils import ClusterHead, LocalClusterHead
    from oracle import heads as OH
    out = {}
    for head_type in ("linear", "mlp"):
        for normalize in (False, True):
            tag = f"{head_type}_norm{int(normalize)}"
            enc = ClusterHead(input_dim=32, num_clusters=6, num_subheads=3, head_type=head_type, T=1, normalize=normalize)
            enc.load_state_dict(OH.init_cluster_head(32, 6, 3, head_type, seed=5))
            outs = enc(T(synth.normal(f"enc_{tag}/feat", (5, 32, 6, 6))))
            for i, o in enumerate(outs):
                out[f"enc_{tag}/out{i}"] = o
            dec = LocalClusterHead(input_dim=8, head_type=head_type, num_clusters=6, num_subheads=3, T=1,
                                   normalize=normalize)
            dec.load_state_dict(OH.init_local_cluster_head(8, 6, 3, head_type, seed=6))
            outs = dec(T(synth.normal(f"dec_{tag}/feat", (3, 8, 10, 12))))
            for i, o in enumerate(outs):
                out[f"dec_{tag}/out{i}"] = o
    save("heads", **out)


def gen_unet():
    from contrastyou.arch import UNet
    from oracle import unet as OU
    out = {}
    net = UNet(input_dim=1, num_classes=4)
    sd0 = OU.init_state(1, 4, seed=3)
    net.load_state_dict(sd0)
    assert list(net.state_dict().keys()) == list(sd0.keys())  # same key order as the reference
    x = T(synth.uniform("unet/x64", (3, 1, 64, 64)))
    wgt = T(synth.normal("unet/w64", (3, 4, 64, 64)))
    net.train()
    logits, enc, dec = net(x, return_features=True)
    for name, f in zip(("Conv5", "Conv4", "Conv3", "Conv2", "Conv1"), enc):
        put_fp(out, f"train64/feat/{name}", f)
    for name, f in zip(("Up_conv5", "Up_conv4", "Up_conv3", "Up_conv2"), dec):
        put_fp(out, f"train64/feat/{name}", f)
    (logits * wgt).sum().backward()
    out["train64/logits"] = logits
    for k, p in net.named_parameters():
        put_fp(out, f"train64/grad/{k}", p.grad)
    for k, v in net.state_dict().items():
        if "running" in k or "num_batches" in k:
            out[f"train64/after/{k}"] = v.clone()  # live buffer: snapshot before the net is reused
    net.eval()
    with torch.no_grad():
        out["eval64/logits"] = net(x)
    net.load_state_dict(sd0)
    net.train()
    with torch.no_grad():
        put_fp(out, "train256/logits", net(T(synth.uniform("unet/x256", (2, 1, 256, 256)))))
    save("unet", **out)


def gen_losses():
    from deepclustering2.loss import KL_div
    from deepclustering2.utils import class2one_hot, simplex, one_hot
    from deepclustering2.augment.tensor_augment import TensorRandomFlip
    from deepclustering2.decorator import FixRandomSeed
    out = {}
    logits = T(synth.normal("kl/logits", (3, 4, 16, 16))).requires_grad_(True)
    target = T(synth.integers("kl/target", (3, 16, 16), 4))
    onehot = class2one_hot(target, 4)
    assert onehot.dtype == torch.int64
    kl = KL_div(verbose=False)(logits.softmax(1), onehot)
    (gl,) = torch.autograd.grad(kl, [logits])
    out.update({"kl/onehot": onehot, "kl/loss": kl, "kl/glogits": gl})
    a = T(synth.normal("mse/a", (3, 4, 16, 16))).requires_grad_(True)
    b = T(synth.normal("mse/b", (3, 4, 16, 16)))
    mse = torch.nn.MSELoss()(a.softmax(1), b.softmax(1).detach())
    (ga,) = torch.autograd.grad(mse, [a])
    out.update({"mse/loss": mse, "mse/ga": ga})
    # the `UDARegCriterion.name: kl` form of the consistency term (semi_seg/trainer.py:137,194 + epocher.py:221-224)
    a2 = T(synth.normal("mse/a", (3, 4, 16, 16))).requires_grad_(True)
    klc = KL_div(verbose=False)(a2.softmax(1), b.softmax(1).detach())
    (ga2,) = torch.autograd.grad(klc, [a2])
    out.update({"klc/loss": klc, "klc/ga": ga2})
    # simplex / one_hot truth table incl. the 1e-4 tolerance edge (general.py:176-196)
    base = torch.full((1, 4, 2, 2), 0.25)
    cases, sx, oh = [], [], []
    for delta in (0.0, 5e-5, 9e-5, 1.9e-4, 2.1e-4, 1e-3):
        t = base.clone()
        t[0, 0] += delta
        cases.append(t)
        sx.append(bool(simplex(t)))
        oh.append(bool(one_hot(t)))
    hot = torch.zeros(1, 4, 2, 2)
    hot[0, 1] = 1
    cases.append(hot)
    sx.append(bool(simplex(hot)))
    oh.append(bool(one_hot(hot)))
    out.update({"simplex/cases": torch.stack(cases), "simplex/is_simplex": np.asarray(sx),
                "simplex/is_one_hot": np.asarray(oh)})
    # flip replay: decisions recovered from the reference's output, plus the flipped tensor
    flipper = TensorRandomFlip(axis=[1, 2], threshold=0.8)
    x = torch.arange(4 * 2 * 3 * 5, dtype=torch.float32).view(4, 2, 3, 5)
    for seed in (0, 123, 9999999, 4242):
        random.seed(555)
        before = random.random()
        random.seed(555)
        with FixRandomSeed(seed):
            flipped = torch.stack([flipper(s) for s in x], dim=0)
        assert random.random() == before  # RNG state restored on exit
        dec = []
        for s_in, s_out in zip(x, flipped):
            found = None
            for ch in (False, True):
                for cw in (False, True):
                    t = s_in.flip(1) if ch else s_in
                    t = t.flip(2) if cw else t
                    if torch.equal(t, s_out):
                        found = [ch, cw]
            dec.append(found)
        out[f"flip/seed{seed}/decisions"] = np.asarray(dec)
        out[f"flip/seed{seed}/out"] = flipped
    save("losses", **out)


def gen_meters_sched():
    from deepclustering2.meters2 import UniversalDice, AverageValueMeter
    from deepclustering2.schedulers import GradualWarmupScheduler
    out = {}
    meter = UniversalDice(4, report_axises=[1, 2, 3])
    groups = []
    for it in range(3):
        p = T(synth.integers(f"dice/pred{it}", (4, 12, 12), 4))
        t = T(synth.integers(f"dice/target{it}", (4, 12, 12), 4))
        grp = [f"patient{(it * 4 + i) % 2:03d}_00" for i in range(4)]
        meter.add(p, t, group_name=grp)
        groups.append(grp)
    summ = meter.summary()
    out["dice/groups"] = np.asarray(groups)
    out["dice/keys"] = np.asarray(list(summ.keys()))
    out["dice/values"] = np.asarray([float(v) for v in summ.values()])
    avg = AverageValueMeter()
    seq = [0.5, 1.25, -0.75, 3.0, 2.0]
    means = []
    for v in seq:
        avg.add(v)
        means.append(float(avg.summary()["mean"]))
    out["avg/seq"], out["avg/means"] = np.asarray(seq), np.asarray(means)
    # lr schedule exactly as SemiTrainer._init_scheduler builds it (trainer.py:52-65), stepped per epoch
    for max_epoch, warm, mult in ((100, 10, 400), (30, 5, 300)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=1e-7, weight_decay=1e-5)
        cos = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=max_epoch - warm, eta_min=1e-7)
        sched = GradualWarmupScheduler(opt, mult, total_epoch=warm, after_scheduler=cos)
        lrs = []
        for e in range(max_epoch):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
        out[f"sched_e{max_epoch}_w{warm}_m{mult}/lrs"] = np.asarray(lrs, dtype=np.float64)
    save("meters_sched", **out)


STEP_SHAPE = dict(H=64, LB=2, UB=3, NB=2, lr=1e-3, wd=1e-5, cons_weight=5.0, iic_weight=0.1)


def step_inputs(mode):
    """Synthetic initial state + batches of the full-step golden (shared with the tests)."""
    from oracle import unet as OU, heads as OH
    H, LB, UB, NB = (STEP_SHAPE[k] for k in ("H", "LB", "UB", "NB"))
    model_sd = OU.init_state(1, 4, seed=9)
    heads = {"Conv5": OH.init_cluster_head(256, 20, 5, "linear", seed=10),
             "Up_conv3": OH.init_local_cluster_head(32, 20, 5, "linear", seed=11),
             "Up_conv2": OH.init_local_cluster_head(16, 20, 5, "linear", seed=12)}
    lab = [(T(synth.uniform(f"step/{mode}/lab{i}", (LB, 1, H, H))), T(synth.integers(f"step/{mode}/tgt{i}", (LB, 1, H, H), 4)))
           for i in range(NB)]
    unl = [T(synth.uniform(f"step/{mode}/unl{i}", (UB, 1, H, H))) for i in range(NB)]
    return model_sd, heads, lab, unl


def gen_step():
    import yaml
    from itertools import chain
    from contrastyou.arch import UNet
    from deepclustering2.loss import KL_div
    from semi_seg._utils import ProjectorWrapper, IICLossWrapper
    from semi_seg.epocher import UDAIICEpocher, TrainEpocher, UDATrainEpocher, IICTrainEpocher
    import semi_seg.epocher as ref_epocher
    H, LB, UB, NB = (STEP_SHAPE[k] for k in ("H", "LB", "UB", "NB"))
    out = {}

    class RecordingAdam(torch.optim.Adam):
        def __init__(self, named, **kw):
            named = list(named)
            self._names = [n for n, _ in named]
            super().__init__([p for _, p in named], **kw)
            self.grad_log = []

        def step(self, closure=None):
            ps = self.param_groups[0]["params"]
            self.grad_log.append({n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
                                  for n, p in zip(self._names, ps)})
            return super().step(closure)

    # udaiic / partial: BASELINE configs; uda / iic: the SURVEY 8(f-4) trainers (semi_seg/epocher.py:200-284) on the same kind of state
    for mode in ("udaiic", "partial", "uda", "iic"):
        cfg = yaml.safe_load(open(os.path.join(REF, "config", "semi.yaml")))
        fn = cfg["Trainer"]["feature_names"]
        fi = [float(v) for v in cfg["Trainer"]["feature_importance"]]
        fi = [v / sum(fi) for v in fi]
        model_sd, heads, lab, unl = step_inputs(mode)
        model = UNet(**cfg["Arch"])
        model.load_state_dict(model_sd)
        pw = ProjectorWrapper()
        pw.init_encoder(feature_names=fn, **cfg["IICRegParameters"]["EncoderParams"])
        pw.init_decoder(feature_names=fn, **cfg["IICRegParameters"]["DecoderParams"])
        pw._encoder_projectors["Conv5"].load_state_dict(heads["Conv5"])
        pw._decoder_projectors["Up_conv3"].load_state_dict(heads["Up_conv3"])
        pw._decoder_projectors["Up_conv2"].load_state_dict(heads["Up_conv2"])
        out[f"{mode}/proj_keys"] = np.asarray(list(pw.state_dict().keys()))
        lw = IICLossWrapper(feature_names=fn, **cfg["IICRegParameters"]["LossParams"])

        def loader(imgs, tgts, B):
            for img, tgt in zip(imgs, tgts):
                yield [[[img, tgt], [img.clone(), tgt.clone()]], [f"patient{i:03d}_00_{i}" for i in range(B)],
                       ["0"] * B, [f"patient{i:03d}_00" for i in range(B)]]

        lab_loader = loader([a for a, _ in lab], [b for _, b in lab], LB)
        unl_loader = loader(unl, [torch.zeros(UB, 1, H, H, dtype=torch.long)] * NB, UB)
        # lr large enough that the update is visible in fp32 (the yaml's 1e-7 barely moves weights)
        named = chain(model.named_parameters(), ((f"proj/{n}", p) for n, p in pw.named_parameters())) \
            if mode in ("udaiic", "iic") else model.named_parameters()
        opt = RecordingAdam(named, lr=STEP_SHAPE["lr"], weight_decay=STEP_SHAPE["wd"])
        seeds = []
        real_randint = random.randint

        def spy(a, b):
            v = real_randint(a, b)
            seeds.append(v)
            return v

        ref_epocher.random.randint = spy
        random.seed(1234)
        try:
            if mode == "udaiic":
                ep = UDAIICEpocher(model, pw, opt, lab_loader, unl_loader, KL_div(verbose=False),
                                   torch.nn.MSELoss(), lw, num_batches=NB, cur_epoch=0, device="cpu",
                                   feature_position=fn, feature_importance=fi,
                                   cons_weight=STEP_SHAPE["cons_weight"], iic_weight=STEP_SHAPE["iic_weight"])
            elif mode == "uda":        # trainer.py:132-147: reg_weight = UDARegCriterion.weight
                ep = UDATrainEpocher(model, opt, lab_loader, unl_loader, KL_div(verbose=False), torch.nn.MSELoss(),
                                     STEP_SHAPE["cons_weight"], NB, 0, "cpu", feature_position=fn, feature_importance=fi)
            elif mode == "iic":        # trainer.py:150-184: reg_weight = IICRegParameters.weight
                ep = IICTrainEpocher(model, pw, opt, lab_loader, unl_loader, KL_div(verbose=False), lw,
                                     STEP_SHAPE["iic_weight"], NB, 0, "cpu", feature_position=fn, feature_importance=fi)
            else:
                ep = TrainEpocher(model, opt, lab_loader, unl_loader, KL_div(verbose=False), 0, NB, 0, "cpu",
                                  feature_position=fn, feature_importance=fi)
            res = ep.run()
        finally:
            ref_epocher.random.randint = real_randint
        out[f"{mode}/seeds"] = np.asarray(seeds, dtype=np.int64)
        flat = {}
        for k, v in res.items():
            for kk, vv in dict(v).items():
                flat[f"{k}/{kk}"] = float(vv)
        out[f"{mode}/meter_keys"] = np.asarray(list(flat.keys()))
        out[f"{mode}/meter_values"] = np.asarray(list(flat.values()), dtype=np.float64)
        out[f"{mode}/param_names"] = np.asarray(list(opt.grad_log[0].keys()))
        for n, gr in opt.grad_log[0].items():
            put_fp(out, f"{mode}/grad_step1/{n}", gr)
        for k, v in model.state_dict().items():
            put_fp(out, f"{mode}/model_after/{k}", v)
        if mode in ("udaiic", "iic"):
            for k, v in pw.state_dict().items():
                put_fp(out, f"{mode}/proj_after/{k}", v)
        out[f"{mode}/feature_importance"] = np.asarray(fi)
    save("step", **out)


_tree = synth.tree_lines


def gen_trainer_io():
    """what the REFERENCE trainers write -- the checkpoint key tree (whl:trainer/_io.py:51-60: every attribute with
    a state_dict + `_buffers`), the `storage.csv` header / index (whl:meters2/storage_interface.py:48-113) and the `config.yaml`
    keys -- after two tiny CPU epochs of each of the four trainers, plus the key tree right after `init()`."""
    import yaml
    import tensorboardX

    class _NoWriter:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def __getattr__(self, k):      # scalar/figure adders only: the real writer has no state_dict, so neither may this one
            if not k.startswith(("add_", "close", "flush")):
                raise AttributeError(k)
            return lambda *a, **kw: None

    import deepclustering2.writer as ref_writer
    import deepclustering2.trainer._trainer as ref_trainer_mod
    ref_writer.SummaryWriter = ref_trainer_mod.SummaryWriter = tensorboardX.SummaryWriter = _NoWriter
    from contrastyou.arch import UNet
    from deepclustering2.loss import KL_div
    from semi_seg.trainer import trainer_zoos
    H, LB, UB = 32, 1, 1
    out = {}

    def train_loader(tag, B):
        i = 0
        while True:
            img = T(synth.uniform(f"tio/{tag}{i}", (B, 1, H, H)))
            tgt = T(synth.integers(f"tio/{tag}t{i}", (B, 1, H, H), 4))
            yield [[[img, tgt], [img.clone(), tgt.clone()]], [f"patient{j:03d}_00_{j}" for j in range(B)], ["0"] * B,
                   [f"patient{j:03d}_00" for j in range(B)]]
            i += 1

    class _Slices(torch.utils.data.Dataset):      # the reference's EvalEpocher insists on a real DataLoader (epocher.py:40)
        def __init__(self, tag):
            self._tag = tag

        def __len__(self):
            return 4

        def __getitem__(self, i):
            p, k = divmod(i, 2)
            return [T(synth.uniform(f"tio/{self._tag}{i}", (1, H, H))), T(synth.integers(f"tio/{self._tag}t{i}", (1, H, H), 4))], \
                f"patient{p:03d}_00_{k:02d}", "0", f"patient{p:03d}_00"

    def EvalLoader(tag):
        return torch.utils.data.DataLoader(_Slices(tag), batch_size=2, shuffle=False)

    for name in ("partial", "uda", "iic", "udaiic"):
        cfg = yaml.safe_load(open(os.path.join(REF, "config", "semi.yaml")))
        cfg["Trainer"].update(name=name, device="cpu", max_epoch=2, num_batches=1, save_dir=f"golden_{name}")
        cfg["Scheduler"]["warmup_max"] = 1
        trainer_name = cfg["Trainer"].pop("name")
        torch.manual_seed(0)
        trainer = trainer_zoos[trainer_name](
            model=UNet(**cfg["Arch"]), labeled_loader=train_loader("lab", LB), unlabeled_loader=train_loader("unl", UB),
            val_loader=EvalLoader("val"), test_loader=EvalLoader("test"), sup_criterion=KL_div(verbose=False),
            configuration={**cfg, "GITHASH": "none"}, **cfg["Trainer"])
        trainer.init()
        out[f"{name}/tree_after_init"] = np.asarray(sorted(_tree(trainer.state_dict())))
        random.seed(7)
        trainer.start_training()
        run = trainer._save_dir
        out[f"{name}/files"] = np.asarray(sorted(f for f in os.listdir(run) if not f.startswith("tensorboard")))
        ck = torch.load(os.path.join(run, "last.pth"), map_location="cpu", weights_only=False)   # the file this very run wrote
        out[f"{name}/tree_last_pth"] = np.asarray(sorted(_tree(ck)))
        out[f"{name}/buffers"] = np.asarray([f"{k}={type(v).__name__}" for k, v in ck["_buffers"].items()])
        with open(os.path.join(run, "storage.csv")) as f:
            rows = f.read().splitlines()
        out[f"{name}/csv_header"] = np.asarray(rows[0].split(","))
        out[f"{name}/csv_index"] = np.asarray([r.split(",")[0] for r in rows[1:]])
        saved_cfg = yaml.safe_load(open(os.path.join(run, "config.yaml")))
        out[f"{name}/config_yaml_keys"] = np.asarray(sorted(f"{k}.{kk}" if isinstance(v, dict) else k for k, v in saved_cfg.items()
                                                              for kk in (v if isinstance(v, dict) else [None])))
    save("trainer_io", **out)


def main():
    """``python make_golden.py [iic heads unet losses meters_sched step ...]`` -- no arguments = every fixture."""
    scratch = import_reference()
    gens = {"iic": gen_iic, "heads": gen_heads, "unet": gen_unet, "losses": gen_losses, "meters_sched": gen_meters_sched,
            "step": gen_step, "trainer_io": gen_trainer_io}
    try:
        torch.set_num_threads(8)
        for name in (sys.argv[1:] or list(gens)):
            gens[name]()
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
