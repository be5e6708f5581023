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


def main(argv=None):
    cmanager = ConfigManger(Path(PROJECT_PATH) / "config/semi.yaml", argv=argv)
    config = cmanager.config
    set_benchmark(config.get("RandomSeed", 1))
    size = int(config.get("Data", {}).get("size", 256))
    classes = config["Arch"]["num_classes"]
    rank = int(os.environ.get("RANK", 0))
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
    trainer.start_training()
    return trainer


if __name__ == "__main__":
    main()
