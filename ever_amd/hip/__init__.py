"""HIP (gfx950) execution layer: ctypes-bound C-ABI kernels wrapped as autograd functions."""
from . import functional  # noqa: F401
from .functional import HipPathError  # noqa: F401


_OPAQUE = [False]


def compiler_opaque(model=None):
    """Mark every kernel-launching entry point of this package — and, given a model, the forward of every module in it that
    this package defines — as opaque to `torch.compile`: Dynamo runs them eagerly and goes on tracing behind them.  The entry
    points are ctypes calls into hand-written kernels and the modules hand Python-side state from layer to layer on the
    tensors (statistics records, operand scales, gradient slots): there is nothing in them a tracing compiler could fuse, and
    much it cannot follow.  What remains visible to the compiler is the user's own module code around these layers.
    Idempotent; called by Trainer.torch_compile (reference trainer.py:241-243)."""
    import torch
    if model is not None:
        for m in model.modules():
            if type(m).__module__.startswith('ever_amd.') and not getattr(m, '_evk_opaque', False):
                m.forward = torch.compiler.disable(m.forward, recursive=True)
                m._evk_opaque = True
    if _OPAQUE[0]:
        return
    import types
    from . import functional_next
    disable = torch.compiler.disable
    for mod in (functional, functional_next):
        for name, obj in list(vars(mod).items()):
            if name.startswith('_') or not isinstance(obj, types.FunctionType):
                continue
            if getattr(obj, '__module__', None) not in (mod.__name__, 'ever_amd.hip.oplib'):
                continue
            setattr(mod, name, disable(obj, recursive=True))
    _OPAQUE[0] = True
