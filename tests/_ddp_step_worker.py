"""Worker of tests/test_gpu_ddp.py: ONE udaiic train step of the real UDAIICEpocher on cuda:0 with this rank's batch, the flat
gradient captured between FlatBuffers.collect() and the Adam kernel.  With WORLD_SIZE=2 (gloo: two ranks share the one GPU of the
box) the GradReducer all-reduces it from the autograd hooks; with WORLD_SIZE unset it is the plain single-process step.

    python tests/_ddp_step_worker.py <data_rank> <out.pt> [dtype]
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mi-based-regularized-semi-supervised-segmentation_amd")]
os.environ.setdefault("MISEG_PROGRESS", "0")

import torch  # noqa: E402

import bench  # noqa: E402
from miseg_amd import _cabi, ddp, ops, unet_ops  # noqa: E402


def main():
    data_rank, out, dtype = int(sys.argv[1]), sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "float32")
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1       # > 4: the launch tape records iteration 4 and replays the rest (MISEG_TAPE=0: all eager)
    _cabi.lib()
    ops.set_mi_precision("fp32" if dtype == "float32" else "bf16x3")
    distributed = ddp.init_from_env()                 # MISEG_DDP_BACKEND=gloo from the one-GPU test, RCCL otherwise
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))     # gloo test: both ranks on cuda:0; RCCL test: one GPU each
    ep, opt = bench.build_step(dev, 2, 3, 64, dtype, data_rank)     # weights: seed 0 on every rank; data: seeded by data_rank
    drv = bench.StepDriver(ep)
    if distributed:
        opt.flat.ensure()
        ep._reducer = ddp.GradReducer(opt.flat, num_buckets=3)
    grabbed, real = [], unet_ops.adam_step

    def spy(param, grad, *a, **k):
        if not grabbed:                 # the first iteration's gradient only (a clone in a later one would be device work outside the
            grabbed.append(grad.detach().clone())     # library, and the launch tape would refuse to record that iteration)
        return real(param, grad, *a, **k)

    unet_ops.adam_step = spy
    random.seed(4321)          # the flip seed draw: the same on every rank and in the single-process runs
    for _ in range(steps):
        drv.step()
    drv.close()
    torch.cuda.synchronize()
    tape = ep._step_tape
    torch.save({"grad": grabbed[0].cpu(), "param_after": opt.flat.flat_param.detach().cpu(),
                "tape_replays": 0 if tape is None else tape.replays, "tape_refused": None if tape is None else tape.disabled,
                "tape_host_calls": 0 if tape is None else len(tape.host_calls),
                "streams": len(getattr(ep._reducer, "producer_streams", [])) if distributed else 0,
                "buckets": len(ep._reducer.buckets) if distributed else 0}, out)
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
