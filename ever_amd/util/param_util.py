"""Parameter freezing / counting helpers used by the encoder (reference ever/util/param_util.py)."""
import numpy as np


def freeze_params(module):
    for p in module.parameters():
        p.requires_grad = False


def freeze_modules(module, specific_class=None):
    for m in module.modules():
        if specific_class is None or isinstance(m, specific_class):
            freeze_params(m)


def count_model_parameters(module, logger=None):
    n = int(sum(np.prod(p.shape) for p in module.parameters()))
    if logger is not None:
        logger.info('# parameters: {} M'.format(round(n / 1e6, 3)))
    return n


def trainable_parameters(module, logger=None):
    n = int(sum(np.prod(p.shape) for p in module.parameters() if p.requires_grad))
    if logger is not None:
        logger.info('# trainable parameters: {} M'.format(round(n / 1e6, 3)))
    return n
