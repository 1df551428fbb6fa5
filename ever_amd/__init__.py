"""ever_amd: MI355X-native engine behind EVer's ERModule / registry / config API.

`import ever_amd as er` mirrors `import ever as er` for the hot path: er.registry, er.ERModule,
er.trainer, er.module, er.builder, er.config ...  `install_as_ever()` additionally aliases the
package as `ever` so unmodified user projects (`import ever as er`) resolve to this engine.
"""
import os
import sys

# Kernel arguments in device memory: a training step is ~800 launches, and with the arguments fetched from host memory each
# one starts later (517.9 vs 533.4 tiles/s on the FarSeg-R50 step).  The runtime's default on gfx950 is already 1; this only
# keeps an environment that says nothing from depending on it.  (Read by the HIP runtime at its first call.)
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

__version__ = '0.1.0'

from .core import builder, config, registry  # noqa: E402
from .core.config import AttrDict  # noqa: E402
from .core.device import auto_device  # noqa: E402
from .core.logger import info  # noqa: E402
from .core.to import to_device, to_tensor  # noqa: E402
from .interface import (Callback, ConfigurableMixin, ERDataLoader, ERDataset, ERModule,  # noqa: E402
                        LearningRateBase)
from . import api, data, magic, metric, module, opt, trainer  # noqa: E402,F401
from .core.launcher import Launcher  # noqa: E402


def install_as_ever():
    """Alias this package (and its sub-packages) as `ever` in sys.modules."""
    this = sys.modules[__name__]
    sys.modules.setdefault('ever', this)
    for name, mod in list(sys.modules.items()):
        if name.startswith(__name__ + '.'):
            sys.modules.setdefault('ever' + name[len(__name__):], mod)
    return this
