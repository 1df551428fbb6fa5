"""Config mixin (API of reference ever/interface/configurable.py:5-36): defaults first, then the
user's dict merged recursively on top; exposed as `.config` / `.cfg`."""
from ..core.config import AttrDict


class ConfigurableMixin:
    def __init__(self, config):
        self._cfg = AttrDict()
        self.set_default_config()
        self._cfg.update(config)

    def set_default_config(self):
        raise NotImplementedError

    @property
    def config(self):
        return self._cfg

    @property
    def cfg(self):
        return self._cfg
