"""config -> object factories (API of reference ever/core/builder.py:6-62)."""
from . import registry

__all__ = ['make_dataloader', 'make_optimizer', 'make_learningrate', 'make_model', 'make_callback']


def _lookup(table, name):
    if name not in table:
        raise ValueError('{} is not support now.'.format(name))
    return table[name]


def make_callback(config):
    return _lookup(registry.CALLBACK, config['type'])(**config['params'])


def make_optimizer(config, params):
    opt = _lookup(registry.OPT, config['type'])(params=params, **config['params'])
    opt.er_config = config  # read back by ERModule.clip_grad (grad_clip)
    return opt


def make_learningrate(config):
    return _lookup(registry.LR, config['type'])(**config['params'])


def make_dataloader(config):
    name = config['type']
    if name in registry.DATALOADER:
        return registry.DATALOADER[name](config['params'])
    if name in registry.DATASET:
        return registry.DATASET[name](config['params']).to_dataloader()
    raise ValueError('{} is not support now.'.format(name))


def make_model(config):
    """ERModule subclasses get the params dict (then `init_from_weight_file`), plain nn.Modules get
    **params — reference builder.py:47-62."""
    from torch.nn import Module
    from ..interface import ERModule
    name = config['type']
    if name not in registry.MODEL:
        raise ValueError('{} is not support now. This model seems not to be registered via '
                         '@er.registry.MODEL.register()'.format(name))
    cls = registry.MODEL[name]
    if inspect_is_subclass(cls, ERModule):
        model = cls(config['params'])
        model.init_from_weight_file()
        return model
    if inspect_is_subclass(cls, Module):
        return cls(**config['params'])
    raise ValueError(f'unsupported model class: {cls}')


def inspect_is_subclass(obj, base):
    return isinstance(obj, type) and issubclass(obj, base)
