#!/usr/bin/env python
"""Benchmark of the udaiic train step (BASELINE.json metric: images/sec, UNet+IIC fwd/bwd, ACDC 256^2).

    python bench.py --gpus N --steps K --warmup W
    (N > 1 either way: under ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`` this
     process IS one of the N ranks; started plainly, it spawns the N ranks itself -- one child process per GPU, before anything
     here touches the GPU, never an exec -- relays rank 0's JSON line and exits non-zero if any rank failed)

A "step" is one full optimisation step of ``UDAIICEpocher`` (one U-Net forward on LB+2*UB slices, supervised KL,
UDA MSE, global+local IIC on three taps x five sub-heads, one backward, fused Adam) on synthetic ACDC-shaped
1x256x256 4-class slices -- BASELINE.json configs[1]: LB=UB=16 per GPU, bf16 compute.  ``value`` = unique loader
slices (LB+UB per rank) per second, whole job.  Rank 0 prints ONE JSON line with, besides the contract fields,
  roofline     : the dominant kernel's algorithmic FLOP/s (HIP events on the launching stream, in the timed region)
  cpu_baseline : the CPU oracle (a port of the reference algorithm) timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(ROOT, "mi-based-regularized-semi-supervised-segmentation_amd")
for p in (ROOT, SRC):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("MISEG_PROGRESS", "0")

import torch  # noqa: E402

FEATURES = ["Conv5", "Up_conv3", "Up_conv2"]
PEAK = {"mfma_bf16": 2500.0, "mfma_f32": 157.3, "hbm": 8000.0}  # TFLOP/s, TFLOP/s, GB/s (MI355X_MICROARCH.md)
# What a register-resident loop of each local-MI instruction mix sustains on this board, [1 ms bursts every 7 ms, continuous] in
# T(FL)OP/s of the instructions issued (profiles/microbench/r03_mfma_dtypes.txt: random operand bits, 2 waves per SIMD, no memory
# traffic).  Round 2 quoted 1 250 here: that loop shuffled accumulators between the register files every iteration.
MICROBENCH_SUSTAINED = {"bf16": [1824, 1999], "bf16x3": [1822, 2006], "f16f8": [2856, 3118]}


def build_step(device, lb, ub, size, dtype, rank, data="synthetic", classes=4, patch=1024):
    from itertools import chain

    from contrastyou.arch import UNet
    from deepclustering2.loss import KL_div
    from deepclustering2.optim import Adam
    from semi_seg._utils import IICLossWrapper, ProjectorWrapper
    from semi_seg.epocher import UDAIICEpocher
    from semi_seg.synthetic import SyntheticPairs
    torch.manual_seed(0)  # identical initial weights on every rank
    model = UNet(input_dim=1, num_classes=classes, compute_dtype=dtype)
    pw = ProjectorWrapper()
    pw.init_encoder(feature_names=FEATURES, num_clusters=20, num_subheads=5, head_types="linear", normalize=False)
    pw.init_decoder(feature_names=FEATURES, num_clusters=20, num_subheads=5, head_types="linear", normalize=False)
    lw = IICLossWrapper(feature_names=FEATURES, paddings=[1, 3], patch_sizes=patch)
    model, pw = model.to(device), pw.to(device)
    opt = Adam(chain(model.parameters(), pw.parameters()), lr=1e-7 * 400, weight_decay=1e-5)
    torch.manual_seed(rank), random.seed(rank)
    if data == "acdc":   # the device input pipeline in the loop: ACDC-format PNG set (synthetic content), 224^2 crops, two fresh views per step
        import contextlib, io, tempfile
        from semi_seg import dataloader_helper as DH
        from semi_seg.synthetic import write_acdc_like
        root = tempfile.mkdtemp(prefix="miseg_acdc_")
        write_acdc_like(root, train_patients=20, val_patients=2, height=256, width=256, seed=rank)
        cfg = {"Data": {"name": "acdc", "labeled_data_ratio": 0.2, "unlabeled_data_ratio": 0.8},
               "LabeledData": {"shuffle": True, "batch_size": lb}, "UnlabeledData": {"shuffle": True, "batch_size": ub}}
        with contextlib.redirect_stdout(io.StringIO()):
            lab, unl, _ = DH.get_dataloaders(cfg, root_dir=root, seed=rank)
    elif data == "host":   # pinned host batches: every step pays the H2D copy (the PCIe-inclusive rate of DESIGN.md section 7; never `value`)
        lab = SyntheticPairs(lb, size, 4, seed=2 * rank, device=None)
        unl = SyntheticPairs(ub, size, 4, seed=2 * rank + 1, device=None)
    else:
        lab = SyntheticPairs(lb, size, classes, seed=2 * rank, device=device)
        unl = SyntheticPairs(ub, size, classes, seed=2 * rank + 1, device=device)
    ep = UDAIICEpocher(model, pw, opt, iter(lab), iter(unl), KL_div(verbose=False), torch.nn.MSELoss(), lw, num_batches=1, cur_epoch=0,
                       device=device, feature_position=FEATURES, feature_importance=[0.5, 0.25, 0.25], cons_weight=5.0, iic_weight=0.1)
    return ep, opt


class StepDriver:
    """Runs UDAIICEpocher steps exactly as ``_run`` does (same code path, incl. the per-iteration meter read-back, which
    -- as in ``_run`` -- is taken after the next iteration is enqueued), but under external timing."""

    def __init__(self, ep):
        from semi_seg._utils import FeatureExtractor
        from semi_seg.epocher import _Pending
        self.ep = ep
        ep._model.train()
        ep.meters = ep._configure_meters(__import__("deepclustering2.meters2", fromlist=["MeterInterface"]).MeterInterface())
        ep._pending = _Pending()
        self._fx = FeatureExtractor(ep._model, ep._feature_position)
        ep._fextractor = self._fx.__enter__()

    def step(self):
        ep = self.ep
        ep._after_step(*ep._step(next(ep._labeled_loader), next(ep._unlabeled_loader)))

    def close(self):
        self.ep._flush_records()
        self._fx.__exit__(None, None, None)


def cpu_baseline(threads):
    """The oracle (CPU port of the reference step) on a bounded sample: one udaiic step at LB=UB=2, 256x256, fp32."""
    from oracle import heads as OH, step as OS, unet as OU
    torch.set_num_threads(threads)
    lb = ub = 2
    g = torch.Generator().manual_seed(0)
    state = OS.StepState(OU.init_state(1, 4, seed=1), {"Conv5": OH.init_cluster_head(256, 20, 5, seed=2),
                                                        "Up_conv3": OH.init_local_cluster_head(32, 20, 5, seed=3),
                                                        "Up_conv2": OH.init_local_cluster_head(16, 20, 5, seed=4)}, lr=4e-5, weight_decay=1e-5)
    lab, tgt = torch.rand(lb, 1, 256, 256, generator=g), torch.randint(0, 4, (lb, 1, 256, 256), generator=g)
    unl = torch.rand(ub, 1, 256, 256, generator=g)
    OS.train_step(state, lab, tgt, unl, seed=122, mode="udaiic")   # untimed: thread pool / allocator warm-up
    t0, steps = time.time(), 0
    while steps < 12 and (steps == 0 or time.time() - t0 < 12.0):   # >= ~12 s of CPU work, bounded
        OS.train_step(state, lab, tgt, unl, seed=123 + steps, mode="udaiic")
        steps += 1
    dt = time.time() - t0
    out = {"value": round(steps * (lb + ub) / dt, 4), "unit": "images/s", "cores": threads, "kind": "port",
           "sample": f"{steps} udaiic train steps of the CPU oracle (oracle/step.py), LB=UB={lb}, 256x256, fp32, {dt:.1f} s after one warm-up step"}
    # the bench's own shape (BASELINE configs[1]: LB=UB=16), ONE step: ~10 s on 16 cores
    big = 16
    lab, tgt = torch.rand(big, 1, 256, 256, generator=g), torch.randint(0, 4, (big, 1, 256, 256), generator=g)
    unl = torch.rand(big, 1, 256, 256, generator=g)
    t0 = time.time()
    OS.train_step(state, lab, tgt, unl, seed=999, mode="udaiic")
    dt2 = time.time() - t0
    out["cfg2_shape"] = {"value": round(2 * big / dt2, 4), "unit": "images/s", "sample": f"1 udaiic train step at LB=UB={big}, 256x256, fp32, {dt2:.1f} s"}
    # BASELINE configs[0] as it stands: the `partial` (supervised-only) trainer, batch 2, num_batches 4 -- the reference's own CPU-runnable case
    state1 = OS.StepState(OU.init_state(1, 4, seed=5), {}, lr=4e-5, weight_decay=1e-5)
    lab, tgt = torch.rand(2, 1, 256, 256, generator=g), torch.randint(0, 4, (2, 1, 256, 256), generator=g)
    unl = torch.rand(2, 1, 256, 256, generator=g)
    OS.train_step(state1, lab, tgt, unl, seed=7, mode="partial")
    t0 = time.time()
    for i in range(4):
        OS.train_step(state1, lab, tgt, unl, seed=8 + i, mode="partial")
    dt1 = time.time() - t0
    out["cfg1"] = {"value": round(4 * 4 / dt1, 4), "unit": "images/s", "sample": f"4 `partial` train steps (num_batches = 4) at LB=UB=2, 256x256, fp32, {dt1:.1f} s after one warm-up step"}
    return out


PMC_FILE = "r04_pmc.json"
PMC_KERNEL = {   # bench tag -> kernel name prefix in profiles/r04_pmc.json (the shipped arithmetic: --mi-precision f16f8)
    "iic_local_bwd[p3]": "local_bwd_f8_kernel<20, 3", "iic_local_bwd[p1]": "local_bwd_rows_kernel<20, 1, 3",
    "iic_local_joint_fwd[p3]": "joint_fwd_px_kernel<3, 3", "iic_local_joint_fwd[p1]": "joint_fwd_px_kernel<1, 3",
}


def pmc_fields(tag, flops_per_call, lib_version, args, path=None):
    """traffic (HBM bytes per launch, FETCH_SIZE + WRITE_SIZE), mfma_busy_frac (SQ_VALU_MFMA_BUSY_CYCLES / all SIMD cycles),
    clock_ghz and ceiling_frac = algorithmic flop / flop of the MFMAs the kernel issues (SQ_INSTS_MFMA x flop per instruction): what
    `frac` would read with the matrix pipe 100 % busy at the clock the peak is quoted for -- the operand split issues extra MFMAs per
    algorithmic product (bf16x3: three; f16f8: one f16 + half a block-scaled fp8 one = the pipe time of two) and tiles pad.  All null unless profiles/r04_pmc.json was taken on this library version, shape and arithmetic."""
    none = {"traffic": None, "mfma_busy_frac": None, "ceiling_frac": None, "clock_ghz": None, "pmc_source": None}
    try:
        pmc = json.load(open(path or os.path.join(ROOT, "profiles", PMC_FILE)))
    except (OSError, ValueError):
        return none
    if pmc.get("lib_version") != lib_version or getattr(args, "mi_precision", None) not in (None, "f16f8") or not (args.lb == 16 and args.ub == 16 and args.size == 256 and args.dtype == "bfloat16" and getattr(args, "config", "cfg2") == "cfg2"):
        return none
    prefix = PMC_KERNEL.get(tag)
    hit = next((v for k, v in pmc.get("kernels", {}).items() if prefix and k.startswith(prefix)), None)
    if hit is None:
        return none
    issued = hit.get("issued_mfma_flop")
    return {"traffic": hit.get("traffic_bytes_factor1"), "mfma_busy_frac": hit.get("mfma_busy_frac"), "clock_ghz": hit.get("clock_ghz"),
            "ceiling_frac": round(flops_per_call / issued, 4) if issued else None,
            "pmc_source": f"profiles/{PMC_FILE} (rocprofv3 --pmc, library version {lib_version}; bash profiles/collect_pmc.sh)"}


def input_pipeline_bench(args):
    """Row 8(f-2): batches from the HBM-resident slices (sampler + parameter draw on the host, one launch per loader)."""
    import tempfile
    import numpy as np
    from semi_seg import dataloader_helper as DH
    from semi_seg.synthetic import write_acdc_like
    from miseg_amd import _cabi
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    root = tempfile.mkdtemp(prefix="miseg_acdc_")
    write_acdc_like(root, train_patients=20, val_patients=2, height=256, width=256, seed=rank)
    cfg = {"Data": {"name": "acdc", "labeled_data_ratio": 0.2, "unlabeled_data_ratio": 0.8},
           "LabeledData": {"shuffle": True, "batch_size": args.lb}, "UnlabeledData": {"shuffle": True, "batch_size": args.ub}}
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        lab, unlab, _ = DH.get_dataloaders(cfg, root_dir=root, seed=rank)
    lab, unlab = iter(lab), iter(unlab)
    _cabi.lib()
    real_call, events = _cabi.call, []

    def timed_call(name, *a, **k):
        if name != "miseg_augment_slices":
            return real_call(name, *a, **k)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        real_call(name, *a, **k)
        e.record()
        events.append((s, e))

    def step():
        return next(lab), next(unlab)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    _cabi.call = timed_call
    import miseg_amd.slices as SL
    SL._cabi.call = timed_call
    t0 = time.perf_counter()
    for _ in range(args.steps):
        keep = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    SL._cabi.call = _cabi.call = real_call
    if world > 1:
        t = torch.tensor([dt], device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    slices = (args.lb + args.ub) * world
    kern_ms = sum(s.elapsed_time(e) for s, e in events) / len(events)
    outs = (args.lb + args.ub)                      # outputs per launch pair -> average launch handles (lb+ub) views
    bytes_per_launch = outs * 224 * 224 * (4 + 8 + 2)   # fp32 image + int64 label written, two u8 gathered, per output pixel
    line = {"metric": "slices/sec (device input pipeline, 2 augmented views per slice)", "value": round(slices * args.steps / dt, 1),
            "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"ACDC-format PNG set resident in HBM, pretrain transform (rot45/flips/crop224/ColorJitter) x2 views, "
                                   f"LB={args.lb} UB={args.ub} per GPU", "parallelism": f"dp{world}"},
            "roofline": {"kernel": "augment_slices", "bound": "hbm", "achieved": round(bytes_per_launch / (kern_ms * 1e-3) / 1e9, 1),
                         "peak": 8000.0, "unit": "GB/s", "frac": round(bytes_per_launch / (kern_ms * 1e-3) / 1e9 / 8000.0, 4),
                         "traffic": None, "avg_ms": round(kern_ms, 4), "launches_per_step": 2,
                         "note": "latency-bound: 32 blocks on 256 CUs; the host parameter draw dominates the step"}}
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import augment as OA
        rng = np.random.default_rng(0)
        img, gt = rng.integers(0, 256, (256, 256), dtype=np.uint8), rng.integers(0, 4, (256, 256), dtype=np.uint8)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.0:
            OA.apply("pretrain", img, gt, n)
            n += 1
        line["cpu_baseline"] = {"value": round(n / (time.perf_counter() - t0), 1), "unit": "slices/s", "cores": 1, "kind": "port",
                                "sample": f"{n} slices through the PIL oracle chain (decode excluded), one core, ~10 s"}
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


def spawn_ranks(n: int, argv) -> int:
    """Plain ``python bench.py --gpus N``: start N child processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    rendezvous on 127.0.0.1), relay rank 0's stdout, wait.  The parent never initialises the GPU and never execs; a failing rank
    ends the others (they would otherwise wait in a collective) and becomes the exit code."""
    import socket
    import subprocess
    import threading
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL across processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=(r == 0)))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
    pump = threading.Thread(target=relay, daemon=True)
    pump.start()
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"[bench] rank {r} exited with code {code}: stopping the other ranks", file=sys.stderr, flush=True)
                for o in live:
                    procs[o].terminate()
        time.sleep(0.05)
    pump.join(timeout=5)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--lb", type=int, default=16)
    ap.add_argument("--ub", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float16", "float32"],
                    help="storage / MFMA operand type of the U-Net and head kernels (float16 = BASELINE configs[4]'s arithmetic: IEEE half, "
                         "dynamic loss scale starting at MISEG_LOSS_SCALE, default 2^14)")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg4"],
                    help="BASELINE.json configs[1] (default: 4 classes, 256^2, whole-map local MI) or configs[3] (8 classes, 512^2, local MI over "
                         "the 7 x 7 grid of overlapping 128^2 patches, +-3 displacement on Up_conv2; --lb / --ub / --size still apply on top)")
    ap.add_argument("--mi-precision", default=None, choices=["fp32", "bf16x3", "f16f8", "bf16"],
                    help="local-MI contraction arithmetic (default: f16f8 with --dtype bfloat16 / float16 -- f16 hi x hi + fp8 cross terms "
                         "where a kernel has that form, the bf16 hi/lo split elsewhere -- and fp32 with --dtype float32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--eager", action="store_true",
                    help="issue every iteration from Python (two threads, ~300 ctypes calls); default on one GPU: after three eager "
                         "iterations the library records its own calls during one more and replays that launch tape from one C call "
                         "per iteration (miseg_amd.tape) -- same launches, same streams, same results, ~1 ms of host time")
    ap.add_argument("--data", default="synthetic", choices=["synthetic", "acdc", "host"],
                    help="synthetic = resident ACDC-shaped tensors (BASELINE metric, default); acdc = batches drawn every step by the "
                         "device input pipeline from an ACDC-format PNG set (224^2 crops, as the reference trains)")
    ap.add_argument("--workload", default="step", choices=["step", "input"],
                    help="step = the udaiic train step (BASELINE metric, default); input = the device-resident input pipeline "
                         "alone (SURVEY.md 8(f-2)): one labeled + one unlabeled batch, two augmented views each")
    ap.add_argument("--spawn", action="store_true", help="start the ranks as child processes even for --gpus 1 (the N > 1 default when "
                                                         "not launched by torch.distributed.run)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        sys.exit(spawn_ranks(args.gpus, [a for a in sys.argv[1:] if a != "--spawn"]))
    if args.workload == "input":
        return input_pipeline_bench(args)
    if args.data == "acdc":
        args.size = 224   # the reference's crop size (semi_seg/augment.py:13)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # stdout carries ONE JSON line (rank 0).  RCCL prints its version banner to stdout when the first communicator comes up: from
    # here on file descriptor 1 is stderr, the line is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    from miseg_amd import _cabi, ddp, ops
    _cabi.lib()
    mi_prec = args.mi_precision or ("f16f8" if args.dtype in ("bfloat16", "float16") else "fp32")
    ops.set_mi_precision(mi_prec)
    distributed = ddp.init_from_env("nccl")

    classes, patch = 4, 1024
    if args.config == "cfg4":        # BASELINE.json configs[3]: 512 x 512, 8 classes, 49 overlapping 128 x 128 patches, pad 3 on Up_conv2
        classes, patch = 8, 128
        if args.size == 256:
            args.size = 512
    ep, opt = build_step(device, args.lb, args.ub, args.size, args.dtype, rank, args.data, classes, patch)
    drv = StepDriver(ep)
    if distributed:
        opt.flat.ensure()
        ep._reducer = ddp.GradReducer(opt.flat, num_buckets=3)

    def barrier():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"built step: world={world} LB=UB={args.lb} {args.size}x{args.size} {args.dtype}")
    # The launch tape (default): iterations 0-2 run eagerly, iteration 3 eagerly while the
    # library records its calls, everything after is replayed.  Per-kernel HIP-event timing of EVERY entry point costs host time (two
    # events per call), so the full table is taken over eager warm-up iterations; inside the timed region only the dominant kernel is
    # timed, live, on the stream it is launched on: by the tape itself when it replays (miseg_tape_time_op), by the ctypes wrapper
    # when the run is eager.
    use_tape = not args.eager      # data-parallel runs too: their all-reduces are host calls between segments of the tape
    if not use_tape:
        ep._TAPE_DEFAULT = False
    eager_warm = min(3, args.warmup) if use_tape else args.warmup
    survey, timer = None, None
    use_timer = rank == 0 and not args.no_kernel_timer
    survey_steps = min(2, eager_warm) if use_timer else 0
    for i in range(args.warmup):
        if use_timer and i == eager_warm - survey_steps:
            torch.cuda.synchronize()
            survey = _cabi.KernelTimer()
            _cabi.TIMER = survey
        if i == eager_warm:
            _cabi.TIMER = None
        drv.step()
        tp = ep._step_tape
        note(f"warmup step {i} done" + (" (launch tape)" if tp is not None and tp.replays else ""))
    _cabi.TIMER = None
    table = []
    timed_ops = []
    if survey is not None and survey.records:
        torch.cuda.synchronize()
        # ranked by calls x SHORTEST duration: an event pair around a kernel of a side stream also counts the time the kernel waited for
        # compute units held by another stream's kernel (a 50 us head forward read 2 ms once) -- the shortest call did not wait
        table = sorted(survey.summary().items(), key=lambda kv: -kv[1]["min_ms"] * kv[1]["calls"])
        timer = _cabi.KernelTimer(only={table[0][0]})      # eager iterations of the timed region (all of them with --eager)
        _cabi.TIMER = timer
    if distributed and getattr(ep, "_reducer", None) is not None:
        ep._reducer.timing = True
        ep._reducer._wait_events = []
    # how long the host is BLOCKED per step (the one wait of an iteration: the previous iteration's scalars, _Pending.wait): the loop
    # time minus this is what the host spends enqueueing a step -- the figure that says whether the host or the GPU limits the rate
    from semi_seg.epocher import _Pending
    blocked = [0.0]
    real_wait = _Pending.wait

    def timed_wait(ticket):
        tb = time.perf_counter()
        out = real_wait(ticket)
        blocked[0] += time.perf_counter() - tb
        return out
    _Pending.wait = staticmethod(timed_wait)
    barrier()
    tape = ep._step_tape
    if tape is not None and tape.handle and table:
        timed_ops = tape.time_tag(table[0][0])             # the recorded iteration's launches of the dominant kernel
    t0 = time.perf_counter()
    for _ in range(args.steps):
        drv.step()
    t_host = time.perf_counter() - t0   # host-side loop time (each step waits for the PREVIOUS step's scalars, so it tracks the GPU)
    t_blocked = blocked[0]
    _Pending.wait = staticmethod(real_wait)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0   # this rank alone: its queue drained, before the barrier that waits for the slowest rank
    barrier()
    dt = time.perf_counter() - t0
    note(f"timed {args.steps} steps in {dt:.3f} s (host loop {t_host:.3f} s)")
    # what each rank saw (N > 1: so that a scaling curve explains itself): its own step time, how long its host spent enqueueing a
    # step, how long its gradient stream waited for the all-reduces after backward had finished (the EXPOSED collective time)
    mine = {"rank": rank, "ms_per_step": round(1000.0 * dt_own / args.steps, 3), "host_loop_ms_per_step": round(1000.0 * t_host / args.steps, 3),
            "host_busy_ms_per_step": round(1000.0 * (t_host - t_blocked) / args.steps, 3),
            "allreduce_exposed_ms_per_step": None}
    if distributed and getattr(ep, "_reducer", None) is not None:
        ex = ep._reducer.exposed_ms()
        mine["allreduce_exposed_ms_per_step"] = None if ex is None else round(ex, 4)
    per_rank = [mine]
    if distributed:
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, mine)
    _cabi.TIMER = None
    drv.close()
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if distributed:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        images = (args.lb + args.ub) * world * args.steps
        out = {
            "metric": "images/sec (UNet+IIC fwd/bwd) ACDC 256^2", "value": round(images / dt, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bfloat16": "bf16", "float16": "f16", "float32": "f32"}[args.dtype],
            "data": {"synthetic": "synthetic", "host": "synthetic, pinned host batches copied to the device every step (PCIe-inclusive)",
                     "acdc": "synthetic ACDC-format PNG set through the device input pipeline (224^2 crops)"}[args.data],
            "config": {"workload": (f"udaiic train step, ACDC-shaped 1x{args.size}x{args.size} 4-class slices, LB=UB={args.lb} per GPU, "
                                    f"taps Conv5/Up_conv3/Up_conv2, K=20 x 5 sub-heads, paddings [1,3] (BASELINE configs[1])") if args.config == "cfg2" else
                                   (f"udaiic train step, synthetic 1x{args.size}x{args.size} 8-class slices, LB=UB={args.lb} per GPU, taps Conv5/Up_conv3/"
                                    f"Up_conv2, K=20 x 5 sub-heads, paddings [1,3], local MI over overlapping 128x128 patches (stride 64: "
                                    f"{(args.size // 64 - 1) ** 2} windows on Up_conv2) (BASELINE configs[3])"),
                       "mi_precision": mi_prec, "global_batch": (args.lb + args.ub) * world, "forward_images_per_step": (args.lb + 2 * args.ub) * world,
                       "parallelism": f"dp{world}",
                       "launch_tape": bool(tape is not None and tape.replays), "tape_ops": (tape.n_ops if tape is not None and tape.handle else None),
                       "tape_replays_in_run": (tape.replays if tape is not None else 0),
                       "tape_refused": (tape.disabled if tape is not None else None)},
            # what the collective library actually saw (1 / null when this is a single process without torch.distributed)
            "rccl_ranks": torch.distributed.get_world_size() if distributed else 1,
            "backend": torch.distributed.get_backend() if distributed else None,
            "per_rank": per_rank,
        }
        if table:
            name, top = table[0]
            ms = [v for op in timed_ops for v in tape.timed_ms(op)] if timed_ops else []
            if timer is not None and timer.records:
                ms += [s.elapsed_time(e) for s, e, _, _ in timer.records[name]]
            if ms:      # live, inside the timed region: HIP events around the kernel on its own stream (tape replays + any eager iterations)
                top = dict(top, avg_ms=sum(ms) / len(ms), min_ms=min(ms), calls=len(ms))
                calls_per_step = len(ms) / args.steps
                timed_in = "timed region (" + ("launch tape replays" if timed_ops else "eager") + ")"
            else:
                calls_per_step, timed_in = top["calls"] / survey_steps, "eager warm-up steps"
            # which matrix pipe the kernel runs on: local-MI follows --mi-precision (bf16x3 = three bf16 MFMAs per
            # algorithmic product, priced against the plain bf16 dense peak), head backward is fp32 MFMA, convs follow --dtype
            mfma_f32 = name.startswith("head_local_bwd") or (name.startswith("iic_local") and mi_prec == "fp32") or \
                (name.startswith("conv3x3") and args.dtype == "float32")
            tf = top["flops_per_call"] / (top["avg_ms"] * 1e-3) / 1e12
            peak = PEAK["mfma_f32"] if mfma_f32 else PEAK["mfma_bf16"]
            out["roofline"] = {"kernel": name, "bound": "mfma", "achieved": round(tf, 3), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(tf / peak, 4), "traffic": None, "avg_ms": round(top["avg_ms"], 4), "calls_per_step": calls_per_step, "events_from": timed_in,
                               "mfma_dtype": "f32" if mfma_f32 else "bf16",
                               "hbm_achieved_GBps": round(top["bytes_per_call"] / (top["avg_ms"] * 1e-3) / 1e9, 1)}
            # Hardware counters of the same kernel / shape from the committed rocprofv3 PMC passes (profiles/collect_pmc.sh; counters
            # cannot be collected in-process).  The file is keyed by kernel name AND library version: numbers taken on another build of
            # the library are ignored (null), never quoted.
            out["roofline"].update(pmc_fields(name, top["flops_per_call"], _cabi.lib().miseg_version(), args))
            if not mfma_f32:
                # what the kernel ISSUES (algorithmic / ceiling_frac, from the committed counter passes) beside what a register-resident
                # loop of the same instruction mix sustains on this board (profiles/microbench/r03_mfma_dtypes.txt, continuous .. 1 ms
                # bursts): two measurements side by side, not a ratio -- they come from different runs
                cf = out["roofline"].get("ceiling_frac")
                out["roofline"]["issued_TFLOPs"] = round(tf / cf, 1) if cf else None
                out["roofline"]["microbench_sustained_TFLOPs"] = MICROBENCH_SUSTAINED.get(mi_prec if name.startswith("iic_local") else "bf16")
            out["kernel_ms_per_step_warmup"] = {k: round(v["total_ms"] / survey_steps, 3) for k, v in table[:int(os.environ.get("MISEG_BENCH_TOP", "10"))]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            out["cpu_baseline"] = cpu_baseline(min(cores, 16))  # the GPU box gives one GPU's share of host cores (16)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
