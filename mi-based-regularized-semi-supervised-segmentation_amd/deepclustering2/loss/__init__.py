from .kl_losses import Entropy, KL_div  # noqa: F401
