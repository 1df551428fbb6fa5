"""Decorator registries (API of reference ever/core/registry.py:46-130): `@MODEL.register()`,
`MODEL.register('name', obj)`, dict access.  User projects register their ERModule / dataloader
classes here and configs refer to them by `type` name."""
import importlib
import importlib.util
import inspect
import logging
import os
import sys

from .dist import is_main_process
from .logger import info

__all__ = ['Registry', 'LR', 'OPT', 'DATALOADER', 'MODEL', 'LOSS', 'OP', 'CALLBACK', 'DATASET', 'register_dir',
           'register_file', 'register_all', 'register_modules', 'register_dataloaders', 'register_callbacks']

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _insert(table, name, obj, override, verbose):
    name = name or obj.__name__
    if is_main_process():
        if not override and name in table:
            logging.warning('{} has been in module_dict.'.format(name))
        if verbose:
            try:
                src = inspect.getfile(obj)
                if not src.startswith(_PKG_ROOT):
                    info(f'{name:<20} is registered from {src}')
            except (TypeError, OSError):
                info(f'<{name}> registered (source unknown)')
    table[name] = obj


class Registry(dict):
    """dict with a `register` method usable as a call (`register(name, obj)`) or as a decorator
    (`@register()` / `@register('alias')`; decorators stack to give several names)."""

    def register(self, module_name=None, module=None, override=False, verbose=True):
        if module is not None:
            _insert(self, module_name, module, override, verbose)
            return None

        def deco(obj):
            _insert(self, module_name, obj, override, verbose)
            return obj

        return deco


def register_dir(dir_name):
    """import every non-underscore .py under ./<dir_name> so their @register decorators run."""
    for root, _dirs, files in os.walk(os.path.join(os.path.curdir, dir_name)):
        if os.path.basename(root).startswith('_'):
            continue
        pkg = '.'.join(root.split(os.path.sep)[1:])
        for f in files:
            if f.endswith('.py') and not f.startswith('_'):
                importlib.import_module(f'{pkg}.{f[:-3]}')


def register_file(file_path):
    spec = importlib.util.spec_from_file_location('ever.custom', file_path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def register_callbacks():
    register_dir('callback')


def register_modules():
    register_dir('module')


def register_dataloaders():
    register_dir('data')


def register_all():
    register_dataloaders()
    register_modules()
    register_callbacks()


LR = Registry()
OPT = Registry()
DATALOADER = Registry()
MODEL = Registry()
LOSS = Registry()
OP = Registry()
CALLBACK = Registry()
DATASET = Registry()
