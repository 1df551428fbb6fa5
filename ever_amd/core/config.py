"""Python-file configs -> attribute dictionaries (API of reference ever/core/config.py:25-122).

`AttrDict` gives item and attribute access, a recursive `update`, and dotted-key overrides from the
command line (`update_from_list`), which is what makes existing EVer config files drop-in.
"""
import copy
import importlib.util
import os
import pickle
import pprint
import sys
import warnings
from ast import literal_eval
from collections import OrderedDict

__all__ = ['import_config', 'AttrDict', 'from_dict', 'to_dict', 'from_pickle']


def _load_py(module_name, path, make_importable=False):
    spec = importlib.util.spec_from_file_location(module_name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if make_importable:
        sys.modules[module_name] = mod
    return mod


def import_config(config_name_or_path, prefix='configs'):
    """`a.b.c` -> ./configs/a/b/c.py ; a `.py` path is loaded directly ; a `.pkl` is unpickled."""
    if config_name_or_path.endswith('.pkl'):
        return from_pickle(config_name_or_path)
    if config_name_or_path.endswith('.py'):
        path = config_name_or_path
    else:
        parts = [prefix] + config_name_or_path.split('.')
        parts[-1] += '.py'
        path = os.path.join(os.path.curdir, *parts)
    return AttrDict.from_dict(_load_py('ever.cfg', path).config)


def from_pickle(path):
    with open(path, 'rb') as f:
        return pickle.load(f)


def to_dict(obj):
    if isinstance(obj, dict):
        return {k: to_dict(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_dict(v) for v in obj]
    return obj


class AttrDict(OrderedDict):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.update(kwargs)

    @staticmethod
    def from_dict(d):
        out = AttrDict()
        out.update(d)
        return out

    def __setitem__(self, key, value):
        OrderedDict.__setitem__(self, key, value)
        OrderedDict.__setattr__(self, key, value)

    __setattr__ = __setitem__

    def update(self, config):
        """Recursive merge: nested dicts merge key-wise, lists of dicts become lists of AttrDict,
        everything else overwrites."""
        for key, val in config.items():
            if isinstance(val, dict):
                cur = self.get(key)
                if not isinstance(cur, dict):
                    cur = AttrDict()
                    self[key] = cur
                cur.update(val)
            elif isinstance(val, list) and all(isinstance(e, dict) for e in val):
                self[key] = [AttrDict.from_dict(e) for e in val]
            else:
                self[key] = val

    def update_from_list(self, str_list):
        """['a.b.0.c', '3', ...] -> self.a.b[0].c = 3 ; values parsed with literal_eval."""
        assert len(str_list) % 2 == 0
        for dotted, raw in zip(str_list[0::2], str_list[1::2]):
            *path, last = dotted.split('.')
            node = self
            for seg in path:
                if isinstance(node, list) and seg.isdigit():
                    seg = int(seg)
                elif isinstance(node, dict) and seg not in node:
                    node[seg] = AttrDict()
                node = node[seg]
            try:
                node[last] = literal_eval(raw)
            except (ValueError, SyntaxError):
                node[last] = raw
                warnings.warn(f'a string {raw} is set to {dotted}')

    def __str__(self):
        return pprint.pformat(self)

    def deepcopy(self):
        return copy.deepcopy(self)

    def to_pickle(self, path):
        with open(path, 'wb') as f:
            pickle.dump(self, f)

    def to_dict(self):
        return to_dict(self)


def from_dict(d):
    return AttrDict.from_dict(d)
