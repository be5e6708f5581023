"""GPU, two processes on the box's one GPU, gloo: the data-parallel path of the REAL train step.

What the design depends on and a toy model cannot show (tests/test_cpu_ddp.py covers the bucketing itself): parameter gradients
are written into their flat-buffer slots from three HIP streams (main backward, IIC branch, weight-gradient side stream), the
reducer's buckets are launched from post-accumulate hooks while the backward is still running, and a bucket must first join every
producer stream (`GradReducer._launch`, `FlatBuffers.collect` -> `join_wgrad_streams`).  Each rank runs one udaiic step of
UDAIICEpocher on its own batch; the all-reduced flat gradient must equal the mean of the two single-process runs on those batches."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_ddp_step_worker.py")


def _run(args, env=None, timeout=600):
    return subprocess.Popen([sys.executable, WORKER] + [str(a) for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_two_rank_step_reduces_to_the_mean_of_the_shard_gradients(tmp_path, dtype):
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    singles = []
    for r in range(2):          # single-process shard runs, one after the other
        p = _run([r, tmp_path / f"single{r}.pt", dtype], env=base)
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-3000:]
        singles.append(torch.load(tmp_path / f"single{r}.pt"))
    port = 32500 + os.getpid() % 2000
    procs = []
    for r in range(2):          # the 2-rank job: both ranks on cuda:0, collectives over gloo
        env = dict(base, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MISEG_DDP_BACKEND="gloo")
        procs.append(_run([r, tmp_path / f"ddp{r}.pt", dtype], env=env))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    d0, d1 = (torch.load(tmp_path / f"ddp{r}.pt") for r in range(2))
    assert d0["buckets"] == 3 and d0["streams"] >= 2                 # the IIC side stream registered itself as a producer
    assert torch.equal(d0["grad"], d1["grad"])                       # identical on both ranks after the all-reduce
    assert torch.equal(d0["param_after"], d1["param_after"])         # ... and so is the optimiser step
    mean = (singles[0]["grad"].double() + singles[1]["grad"].double()) / 2
    got = d0["grad"].double()
    # same kernels, same inputs per rank: the only difference is (a + b) / 2 in fp32 on the host path of gloo
    err = float((got - mean).abs().max())
    scale = float(mean.abs().max())
    assert err <= 2e-6 * scale, (err, scale)
    assert float((singles[0]["grad"] - singles[1]["grad"]).abs().max()) > 1e-3 * scale     # the shards really differ


def test_two_rank_launch_tape_equals_two_rank_eager(tmp_path):
    """The data-parallel iteration under the launch tape: the bucket all-reduces and the wait for them are host calls between the
    tape's segments (miseg_amd.tape.host_call), repeated at every replay.  Seven iterations of two ranks (gloo, both on the one GPU):
    three eager, one recorded, three replayed -- parameters must equal, bit for bit, those of the same job run eagerly throughout."""
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    results = {}
    for mode, tape in (("tape", "1"), ("eager", "0")):
        port = 36500 + (os.getpid() + (7 if tape == "1" else 0)) % 2000
        procs = []
        for r in range(2):
            env = dict(base, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MISEG_DDP_BACKEND="gloo",
                       MISEG_TAPE=tape, MISEG_TAPE_QUIET="0")
            procs.append(_run([r, tmp_path / f"{mode}{r}.pt", "bfloat16", 7], env=env))
        outs = [p.communicate(timeout=900)[0] for p in procs]
        for p, o in zip(procs, outs):
            assert p.returncode == 0, o[-3000:]
        results[mode] = [torch.load(tmp_path / f"{mode}{r}.pt") for r in range(2)]
    t0, t1 = results["tape"]
    assert t0["tape_refused"] is None and t0["tape_replays"] == 3 and t0["tape_host_calls"] == 4, (t0["tape_refused"], t0["tape_replays"], t0["tape_host_calls"])
    assert torch.equal(t0["param_after"], t1["param_after"])
    assert results["eager"][0]["tape_replays"] == 0
    assert torch.equal(t0["param_after"], results["eager"][0]["param_after"])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: the test box has a single GPU (the driver's 8-GPU node runs it)")
def test_two_rank_step_over_rccl(tmp_path):
    """The same two-rank step over the `nccl` backend (= RCCL on ROCm), one GPU per rank: bucketed asynchronous all-reduce (native AVG)
    from the autograd hooks on the collective's own stream, joined by `GradReducer.finish`.  Skips below two visible GPUs."""
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MISEG_DDP_BACKEND")}
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    singles = []
    for r in range(2):
        p = _run([r, tmp_path / f"single{r}.pt", "bfloat16"], env=base)
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-3000:]
        singles.append(torch.load(tmp_path / f"single{r}.pt"))
    port = 34500 + os.getpid() % 2000
    procs = [_run([r, tmp_path / f"rccl{r}.pt", "bfloat16"],
                  env=dict(base, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    d0, d1 = (torch.load(tmp_path / f"rccl{r}.pt") for r in range(2))
    assert d0["buckets"] == 3 and torch.equal(d0["grad"], d1["grad"]) and torch.equal(d0["param_after"], d1["param_after"])
    mean = (singles[0]["grad"].double() + singles[1]["grad"].double()) / 2
    err, scale = float((d0["grad"].double() - mean).abs().max()), float(mean.abs().max())
    assert err <= 2e-6 * scale, (err, scale)
