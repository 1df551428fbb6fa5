from . import learning_rate, optimizer  # noqa: F401  (registers LR / OPT entries)
from .optimizer import FusedAdam, FusedAdamW, FusedSGD  # noqa: F401
