from .acdc_dataset import ACDCDataset, ACDCSemiInterface  # noqa: F401
