"""Entry point with the reference's surface: ``python semi_seg/main.py Trainer.name=udaiic key=value ...``
(ref: semi_seg/main.py:1-45).  ``Data.name=acdc`` reads ``<Data.root or .data>/ACDC_contrast`` through the device-resident
input pipeline (semi_seg/dataloader_helper.py); when the dataset is absent (it cannot be downloaded here)
or ``Data.name=synthetic``, ACDC-shaped synthetic slices are fed instead."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from contrastyou import PROJECT_PATH  # noqa: E402
from contrastyou.arch import UNet  # noqa: E402
from deepclustering2.configparser import ConfigManger  # noqa: E402
from deepclustering2.loss import KL_div  # noqa: E402
from deepclustering2.utils import gethash, set_benchmark  # noqa: E402
from contrastyou import DATA_PATH  # noqa: E402
from semi_seg.dataloader_helper import create_val_loader, get_dataloaders  # noqa: E402
from semi_seg.synthetic import SyntheticEval, SyntheticPairs  # noqa: E402
from semi_seg.trainer import trainer_zoos  # noqa: E402


def _place_rank(config) -> int:
    """One process per GPU (``python -m torch.distributed.run --nproc-per-node N semi_seg/main.py ...``): bind this process to
    its own device BEFORE anything touches the GPU, point ``Trainer.device`` at it and join the RCCL job.  Returns the rank."""
    import torch
    from miseg_amd import ddp
    local = int(os.environ.get("LOCAL_RANK", 0))
    wants_gpu = str(config["Trainer"].get("device", "cpu")).startswith("cuda")
    if int(os.environ.get("WORLD_SIZE", 1)) > 1 or os.environ.get("MISEG_FORCE_DDP", "0") == "1":
        if wants_gpu:
            torch.cuda.set_device(local)
            config["Trainer"]["device"] = f"cuda:{local}"
        ddp.init_from_env("nccl" if wants_gpu else "gloo")     # MISEG_DDP_BACKEND overrides (gloo also moves device tensors)
    return int(os.environ.get("RANK", 0))


def build_trainer(argv=None):
    """Everything of ``main`` up to (not including) the training loop: config, rank placement, loaders, model, trainer,
    optional checkpoint, data-parallel attachment."""
    cmanager = ConfigManger(Path(PROJECT_PATH) / "config/semi.yaml", argv=argv)
    config = cmanager.config
    rank = _place_rank(config)
    set_benchmark(config.get("RandomSeed", 1))     # identical initial weights on every rank; the data seeds below differ per rank
    size = int(config.get("Data", {}).get("size", 256))
    classes = config["Arch"]["num_classes"]
    data_root = config.get("Data", {}).get("root") or DATA_PATH
    data_name = config.get("Data", {}).get("name", "acdc")
    if data_name == "acdc" and (Path(data_root) / "ACDC_contrast").is_dir():
        labeled_loader, unlabeled_loader, test_loader = get_dataloaders(config, root_dir=data_root, seed=rank)
        val_loader = create_val_loader(unlabeled_loader, test_loader)
    else:
        if data_name == "acdc":
            print(f"{data_root}/ACDC_contrast not found: feeding synthetic ACDC-shaped slices")
        labeled_loader = SyntheticPairs(config["LabeledData"]["batch_size"], size, classes, seed=2 * rank)
        unlabeled_loader = SyntheticPairs(config["UnlabeledData"]["batch_size"], size, classes, seed=2 * rank + 1)
        val_loader, test_loader = SyntheticEval(2, 4, size, classes, seed=100), SyntheticEval(2, 4, size, classes, seed=101)
    trainer_name = config["Trainer"].pop("name")
    model = UNet(**config["Arch"])
    trainer = trainer_zoos[trainer_name](
        model=model, labeled_loader=iter(labeled_loader), unlabeled_loader=iter(unlabeled_loader), val_loader=val_loader,
        test_loader=test_loader, sup_criterion=KL_div(), configuration={**cmanager.config, **{"GITHASH": gethash(__file__)}},
        **config["Trainer"])
    trainer.init()
    checkpoint = config.get("Checkpoint", None)
    if checkpoint is not None:
        trainer.load_state_dict_from_path(checkpoint, strict=False)
    trainer.attach_data_parallel()      # no-op for a single process; else bucketed RCCL all-reduce of the flat gradient
    return trainer


def main(argv=None):
    trainer = build_trainer(argv)
    try:
        trainer.start_training()
    finally:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
    return trainer


if __name__ == "__main__":
    main()
