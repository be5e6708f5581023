"""Reference-compatible ``contrastyou`` namespace backed by the MI355X kernels (``miseg_amd``).

Only the parts on the semi-supervised train-step hot path are provided: the
U-Net, the IIC losses, the cluster heads and the small helpers the epochers use.
"""
from pathlib import Path

PROJECT_PATH = str(Path(__file__).parents[1])
DATA_PATH = str(Path(PROJECT_PATH) / ".data")
CONFIG_PATH = str(Path(PROJECT_PATH, "config"))
