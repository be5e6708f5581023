"""CPU, world_size 2, gloo: the data-parallel gradient reducer (miseg_amd.ddp.GradReducer) -- bucketed async
all-reduce of the flat gradient launched from autograd hooks -- equals a single-process run on the
concatenated batch shards (mean of per-rank gradients), including a parameter that gets no gradient."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mi-based-regularized-semi-supervised-segmentation_amd")


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                               torch.nn.Linear(16, 3), torch.nn.Linear(3, 3))  # the last layer is never used -> no grad


def _loss(model, x, y):
    h = model[4](model[3](model[2](model[1](model[0](x)))))
    return ((h - y) ** 2).mean()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, SRC)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from miseg_amd import ddp
    from miseg_amd.flat import FlatBuffers
    assert ddp.init_from_env("gloo")
    model = _model()
    if rank == 1:  # de-synchronise on purpose: the reducer must broadcast rank 0's parameters
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    flat = FlatBuffers(list(model.parameters()))
    red = ddp.GradReducer(flat, num_buckets=3)
    assert len(red.buckets) >= 2 and red.buckets[0][3] == flat.total and red.buckets[-1][2] == 0
    g = torch.Generator().manual_seed(100 + rank)
    x, y = torch.randn(5, 6, generator=g), torch.randn(5, 3, generator=g)
    for it in range(3):  # state resets correctly; from the second pass on only the learned trigger hooks are left
        flat.zero_grad()
        red.prepare()
        _loss(model, x, y).backward()
        red.finish()
        assert red._triggers_only and sum(1 for i in red._last_in_bucket if i is not None) == len(red.buckets)
    # a pass in which the gradients arrive in ANOTHER order (the last layer's output is not used: its bucket's trigger never fires,
    # an earlier bucket's trigger fires while ... ) must still reduce everything: finish() flushes what the triggers left
    flat.zero_grad()
    red.prepare()
    h = model[2](model[1](model[0](x)))
    (h ** 2).mean().backward()
    red.finish()
    partial = flat.flat_grad.clone()
    ref_partial = [torch.zeros_like(partial) for _ in range(world)]
    dist.all_gather(ref_partial, partial)
    assert all(torch.equal(ref_partial[0], t) for t in ref_partial)        # every rank holds the same (averaged) gradient
    assert float(partial.abs().max()) > 0
    flat.zero_grad()
    red.prepare()
    _loss(model, x, y).backward()
    red.finish()
    torch.save({"grad": flat.flat_grad.clone(), "param": flat.flat_param.clone()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_grad_reducer_matches_single_process(tmp_path):
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"r{r}.pt") for r in range(world))
    torch.testing.assert_close(r0["grad"], r1["grad"], rtol=0, atol=0)       # identical on every rank
    torch.testing.assert_close(r0["param"], r1["param"], rtol=0, atol=0)     # broadcast from rank 0
    # single-process reference: mean over ranks of the per-rank gradients
    sys.path.insert(0, SRC)
    from miseg_amd.flat import FlatBuffers
    grads = []
    for rank in range(world):
        model = _model()
        flat = FlatBuffers(list(model.parameters()))
        flat.zero_grad()
        g = torch.Generator().manual_seed(100 + rank)
        x, y = torch.randn(5, 6, generator=g), torch.randn(5, 3, generator=g)
        _loss(model, x, y).backward()
        flat.collect()   # gradients autograd parked outside the flat buffer -> their slots; unused parameters -> zeros
        grads.append(flat.flat_grad.clone())
    torch.testing.assert_close(r0["grad"], (grads[0] + grads[1]) / 2, rtol=1e-6, atol=1e-7)
    assert float(r0["grad"][-12:].abs().max()) == 0.0   # the unused layer's slot stays exactly zero


# ------------------------------------------------------------------------------------------ through the entry point
def _main_worker(rank, world, port, out_dir):
    """``semi_seg/main.py``'s own wiring (build_trainer = main minus the loop) in a 2-rank gloo job on CPU tensors: rank
    placement, process group, trainer construction, GradReducer attachment, rank-0-only run directory.  The kernels are GPU-only,
    so the gradient that travels is produced by a surrogate loss on the real parameters (each rank scales it differently)."""
    sys.path.insert(0, SRC)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MISEG_PROGRESS="0")
    from semi_seg.main import build_trainer
    save_dir = os.path.join(out_dir, "run")      # the SAME directory on both ranks, as in a real job
    trainer = build_trainer(["Trainer.name=udaiic", "Trainer.device=cpu", f"Trainer.save_dir={save_dir}", "Trainer.max_epoch=1",
                             "Trainer.num_batches=1", "Data.name=synthetic", "Data.size=32", "LabeledData.batch_size=1",
                             "UnlabeledData.batch_size=1"])
    assert dist.is_initialized() and dist.get_world_size() == world and dist.get_backend() == "gloo"
    red = trainer._grad_reducer
    assert red is not None and red.world == world and trainer.is_writer == (rank == 0)
    flat = trainer._optimizer.flat
    assert red.flat is flat and flat.flat_param.numel() > 2_000_000      # U-Net + the 15 projector sub-heads in ONE buffer
    for _ in range(2):
        trainer._optimizer.zero_grad()
        red.prepare()
        params = list(trainer._model.parameters()) + list(trainer._projector_wrappers.parameters())
        sum(((rank + 1.0) * 0.5 * (p ** 2).sum()) for p in params).backward()     # d/dp = (rank + 1) p
        red.finish()
    # the trainer's own logging / checkpoint path: every rank calls it, only the writer touches the directory
    from deepclustering2.meters2 import StorageIncomeDict
    trainer._cur_epoch = 0
    with trainer._writer_context() as trainer._writer:
        trainer._log_epoch(StorageIncomeDict(tra={"loss": {"mean": 1.0}}, val={"loss": {"mean": 2.0}}, test={"loss": {"mean": 3.0}}), 0.5)
    assert trainer._best_score == 0.5
    torch.save({"grad": flat.flat_grad.clone(), "param": flat.flat_param.clone()}, os.path.join(out_dir, f"m{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_main_entry_point_wires_data_parallel(tmp_path):
    world, port = 2, 31500 + os.getpid() % 2000
    mp.spawn(_main_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"m{r}.pt") for r in range(world))
    torch.testing.assert_close(r0["param"], r1["param"], rtol=0, atol=0)      # same seed + broadcast
    torch.testing.assert_close(r0["grad"], r1["grad"], rtol=0, atol=0)
    torch.testing.assert_close(r0["grad"], 1.5 * r0["param"], rtol=1e-6, atol=1e-9)    # mean of (1 p, 2 p)
    run = tmp_path / "run"
    assert sorted(p.name for p in run.iterdir() if p.suffix in (".pth", ".yaml", ".csv")) == ["best.pth", "config.yaml", "last.pth", "storage.csv"]
