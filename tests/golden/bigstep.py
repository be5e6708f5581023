"""The two whole-step oracle runs that take minutes of CPU (BASELINE configs[1] at its own size, configs[3] reduced), cached.

tests/test_gpu_step.py::test_whole_step_at_the_bench_shape_matches_the_oracle and ::test_cfg4_step_matches_the_oracle hold the HIP step
against ``oracle.step.train_step`` on deterministic synthetic inputs (tests/golden/synth.py) -- in fp32 and with the U-Net's bf16
rounding points emulated.  Those four oracle steps cost ~4 minutes of host time on the GPU box, a third of the GPU suite's budget, and
never change unless the oracle or the input generator does.  So their outputs (every scalar of the step, the gradients of the logits
layer and of the head parameters) are kept in ``bigstep_oracle.npz`` next to this file, written by

    python tests/golden/bigstep.py            # ~6 min on 8 cores; needs neither the reference nor a GPU

together with a fingerprint of the sources that determine them (oracle/*.py, synth.py, this file).  ``result(...)`` returns the cached
arrays only while that fingerprint still matches; otherwise it recomputes -- a stale cache can cost time, never hide a change.
The oracle itself is pinned to the reference by tests/test_oracle_golden.py; this file caches ORACLE outputs, not reference outputs."""
import hashlib
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import synth  # noqa: E402
from oracle import heads as OH  # noqa: E402
from oracle import step as OS  # noqa: E402
from oracle import unet as OU  # noqa: E402

CACHE = os.path.join(HERE, "bigstep_oracle.npz")
T = torch.from_numpy
FEATURES = ["Conv5", "Up_conv3", "Up_conv2"]


def fingerprint() -> str:
    h = hashlib.sha1()
    files = sorted(os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith(".py"))
    for f in files + [os.path.join(HERE, "synth.py"), os.path.abspath(__file__)]:
        h.update(open(f, "rb").read())
    return h.hexdigest()


def cfg2_inputs():
    """BASELINE configs[1] at its own size: LB = UB = 16, 256 x 256, default taps / heads / paddings."""
    H, LB, UB = 256, 16, 16
    heads = {"Conv5": OH.init_cluster_head(256, 20, 5, "linear", seed=41), "Up_conv3": OH.init_local_cluster_head(32, 20, 5, "linear", seed=42),
             "Up_conv2": OH.init_local_cluster_head(16, 20, 5, "linear", seed=43)}
    limg, ltgt = T(synth.uniform("cfg2step/lab", (LB, 1, H, H))), T(synth.integers("cfg2step/tgt", (LB, 1, H, H), 4))
    uimg = T(synth.uniform("cfg2step/unl", (UB, 1, H, H)))
    random.seed(2468)
    seed = random.randint(0, int(1e7))               # the flip seed the epocher draws for iteration 1 (ref epocher.py:146)
    return dict(heads=heads, limg=limg, ltgt=ltgt, uimg=uimg, seed=seed, unet_seed=40, classes=4)


def cfg4_inputs():
    """BASELINE configs[3] reduced to what the oracle finishes in a minute: 8 classes, 7 x 7 half-overlapping 64-pixel patches on 256 x 256."""
    H, LB, UB, NC = 256, 1, 1, 8
    heads = {"Conv5": OH.init_cluster_head(256, 20, 5, "linear", seed=51), "Up_conv3": OH.init_local_cluster_head(32, 20, 5, "linear", seed=52),
             "Up_conv2": OH.init_local_cluster_head(16, 20, 5, "linear", seed=53)}
    limg, ltgt = T(synth.uniform("cfg4step/lab", (LB, 1, H, H))), T(synth.integers("cfg4step/tgt", (LB, 1, H, H), NC))
    uimg = T(synth.uniform("cfg4step/unl", (UB, 1, H, H)))
    random.seed(1357)
    seed = random.randint(0, int(1e7))
    return dict(heads=heads, limg=limg, ltgt=ltgt, uimg=uimg, seed=seed, unet_seed=50, classes=NC, patch=64)


def compute(which: str, bf16: bool):
    """(scalars dict, gradients dict) of one oracle udaiic step of configuration ``which`` ('cfg2' | 'cfg4')."""
    inp = cfg2_inputs() if which == "cfg2" else cfg4_inputs()
    state = OS.StepState(OU.init_state(1, inp["classes"], seed=inp["unet_seed"]), inp["heads"], lr=1e-3, weight_decay=1e-5)
    kw = dict(mode="udaiic", feature_importance=[0.5, 0.25, 0.25], cons_weight=5.0, iic_weight=0.1, do_update=False)
    if which == "cfg4":
        kw.update(paddings=[1, 3], patch_sizes=[inp["patch"], inp["patch"]], num_classes=inp["classes"])
    if bf16:
        kw["unet_fn"] = OU.unet_forward_bf16_autograd
    threads = torch.get_num_threads()
    torch.set_num_threads(max(threads, min(16, os.cpu_count() or 1)))
    try:
        sc, grads = OS.train_step(state, inp["limg"], inp["ltgt"], inp["uimg"], inp["seed"], **kw)
    finally:
        torch.set_num_threads(threads)
    keep = {k: v for k, v in grads.items() if "/" in k or k.startswith("DeConv_1x1")}      # what the tests compare: logits layer + heads
    return {k: float(v) for k, v in sc.items() if isinstance(v, (int, float)) or (torch.is_tensor(v) and v.numel() == 1)}, keep


_mem = {}


def result(which: str, bf16: bool):
    """Cached (scalars, gradients) of ``compute(which, bf16)``: from the fixture while its fingerprint matches, else recomputed."""
    key = f"{which}/{'bf16' if bf16 else 'fp32'}"
    if key in _mem:
        return _mem[key]
    if os.path.exists(CACHE):
        z = np.load(CACHE, allow_pickle=False)
        if str(z["fingerprint"]) == fingerprint() and f"{key}/scalar_names" in z.files:
            names = [str(n) for n in z[f"{key}/scalar_names"]]
            sc = dict(zip(names, [float(v) for v in z[f"{key}/scalar_values"]]))
            grads = {str(n): T(z[f"{key}/grad/{n}"].copy()) for n in z[f"{key}/grad_names"]}
            _mem[key] = (sc, grads)
            return _mem[key]
    _mem[key] = compute(which, bf16)
    return _mem[key]


def main():
    out = {"fingerprint": np.array(fingerprint())}
    for which in ("cfg2", "cfg4"):
        for bf16 in (False, True):
            key = f"{which}/{'bf16' if bf16 else 'fp32'}"
            sc, grads = compute(which, bf16)
            out[f"{key}/scalar_names"] = np.array(sorted(sc))
            out[f"{key}/scalar_values"] = np.array([sc[k] for k in sorted(sc)], dtype=np.float64)
            out[f"{key}/grad_names"] = np.array(sorted(grads))
            for n, g in grads.items():
                out[f"{key}/grad/{n}"] = g.detach().numpy().astype(np.float32)
            print(key, {k: round(v, 6) for k, v in sc.items()}, len(grads), "gradient tensors", flush=True)
    np.savez_compressed(CACHE, **out)
    print("wrote", CACHE, os.path.getsize(CACHE), "bytes")


if __name__ == "__main__":
    main()
