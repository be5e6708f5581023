"""``get_dataloaders`` / ``create_val_loader`` with the reference's signatures (semi_seg/dataloader_helper.py:23-109):
patient-level semi-supervised split, infinite random loaders for the labeled / unlabeled sets, one batch per patient for
validation.  The loaders are ``miseg_amd.slices`` device loaders: the dataset is resident in HBM and a batch is one launch,
so ``num_workers`` / ``pin_memory`` have nothing to configure (accepted in the config, unused)."""
from contextlib import contextmanager
from copy import deepcopy

import numpy as np

from contrastyou import DATA_PATH
from contrastyou.dataloader import ACDCDataset, ACDCSemiInterface
from miseg_amd.slices import AugmentedLoader, PatientLoader
from semi_seg.augment import ACDCStrongTransforms

dataset_zoos = {"acdc": ACDCSemiInterface}
augment_zoos = {"acdc": ACDCStrongTransforms}


def get_dataloaders(config, group_val_patient=True, root_dir=None, device=None, seed=0):
    _config = deepcopy(config)
    dataset_name = _config["Data"].pop("name", "acdc")
    assert dataset_name in dataset_zoos.keys(), config["Data"]
    augment = augment_zoos[dataset_name]
    manager = dataset_zoos[dataset_name](root_dir=root_dir or DATA_PATH, labeled_data_ratio=config["Data"]["labeled_data_ratio"],
                                         unlabeled_data_ratio=config["Data"]["unlabeled_data_ratio"], device=device)
    label_set, unlabel_set, val_set = manager._create_semi_supervised_datasets(  # noqa
        labeled_transform=augment.pretrain, unlabeled_transform=augment.pretrain, val_transform=augment.val)
    labeled_loader = AugmentedLoader(label_set, batch_size=config["LabeledData"]["batch_size"],
                                     shuffle=config["LabeledData"]["shuffle"], seed=2 * seed)
    unlabeled_loader = AugmentedLoader(unlabel_set, batch_size=config["UnlabeledData"]["batch_size"],
                                       shuffle=config["UnlabeledData"]["shuffle"], seed=2 * seed + 1)
    assert group_val_patient, "slice-batched validation is not on the reference's default path"
    return labeled_loader, unlabeled_loader, PatientLoader(val_set)


@contextmanager
def fix_numpy_seed(seed: int = 1):
    previous_state = np.random.get_state()
    np.random.seed(seed)
    yield
    np.random.set_state(previous_state)


def create_val_loader(unlabeled_loader, test_loader):
    """Five patients of the unlabeled set, drawn with numpy seed 1, evaluated with the test transform
    (dataloader_helper.py:82-109)."""
    unlabeled_dataset: ACDCDataset = unlabeled_loader.dataset
    patient_group = sorted(unlabeled_dataset.show_group_set())
    with fix_numpy_seed(1):
        val_patient = np.random.permutation(patient_group)[:5]
    files = unlabeled_dataset.get_filenames()
    val_dataset = unlabeled_dataset.subset([unlabeled_dataset._get_group(f) in val_patient for f in files])
    val_dataset.set_transform(deepcopy(test_loader.dataset.transform))
    return PatientLoader(val_dataset)
