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
        # per CLASS, not per instance: an instance attribute `forward` bound to the original module is copied by
        # reference by copy.deepcopy (an EMA / SWA copy would silently run the ORIGINAL's weights) and breaks torch.save(model)
        for m in model.modules():
            cls = type(m)
            if cls.__module__.startswith('ever_amd.') and not cls.__dict__.get('_evk_opaque', False):
                cls.forward = torch.compiler.disable(cls.forward, recursive=True)
                cls._evk_opaque = True
        if not getattr(model, '_evk_opaque_logged', False):
            import logging
            logging.getLogger('EVER').info('torch_compile: the built-in HIP layers run eagerly (hand-written kernels behind '
                                           'ctypes: nothing to trace); only user module code around them is compiled')
            try:
                model._evk_opaque_logged = True
            except Exception:
                pass
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
