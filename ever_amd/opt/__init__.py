from . import learning_rate, optimizer  # noqa: F401  (registers LR / OPT entries)
from .optimizer import FusedSGD  # noqa: F401
