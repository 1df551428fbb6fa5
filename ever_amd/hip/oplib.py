"""The inference entry points of hip/functional.py as `torch.library` operators (namespace `ever_amd`).

SURVEY §8 b4 asked for the C-ABI to be exposed as PyTorch-ROCm ops; the training path does that with
`autograd.Function` wrappers around ctypes calls, which a tracer cannot see through (TorchScript records a PythonOp it
cannot serialise).  The reference exports models with `torch.jit.trace` (`ever/api/infer_tool.py:70-74`) and offers
`torch.compile` (`ever/trainer/trainer.py:241-243`), so the no-grad forward of every layer kind is ALSO a dispatcher-level
operator: an opaque custom op with a schema, the same ctypes launch behind it, and a fake (shape-only) implementation.

  * the operators differentiate (register_autograd: the backward re-runs the entry point under autograd — same kernels, same
    gradients; the package's own training path keeps its state-carrying autograd.Function wrappers and never pays that);
  * eager calls do not go through the dispatcher (no cost on the hot path): `traceable()` returns a wrapper that takes the
    operator route only while a trace is being recorded (`torch.jit.trace`, or a compiler's fake-tensor pass);
  * a TorchScript file saved from such a trace holds `ever_amd::conv2d(...)` nodes and the weights as constants; loading it
    needs `import ever_amd` (which registers the operators) and an MI355X — there is still no CPU path.
"""
import threading

import torch

__all__ = ['traceable', 'tracing', 'OPS']

_LIB = torch.library.Library('ever_amd', 'DEF')
_STATE = threading.local()
OPS = {}


def tracing():
    """a TorchScript trace or a compiler's symbolic pass is being recorded on this thread"""
    if getattr(_STATE, 'inside', False):
        return False
    if torch._C._get_tracing_state() is not None:
        return True
    is_compiling = getattr(torch.compiler, 'is_compiling', None)
    return bool(is_compiling and is_compiling())


def traceable(name, schema, fn, adapt, fake, applies=None, impl_fn=None):
    """Register `fn` as ever_amd::<name> with `schema` ("(Tensor x, ...) -> Tensor") and return the function the package
    calls: `fn` itself, except under a trace, where the call is routed through the operator so that the tracer records it.
    impl_fn: what the operator runs on its positional arguments when that is not `fn` itself (other signature).

    adapt(*args, **kwargs) -> the operator's positional arguments (tensors, ints, floats, bools, lists of ints);
    fake(*op_args) -> an empty tensor of the result's shape / strides (shape propagation for torch.compile / export);
    applies(*args, **kwargs) -> False keeps a call on the plain path even under a trace (e.g. training-mode arguments)."""
    _LIB.define(name + schema)

    def impl(*op_args):
        _STATE.inside = True          # nested entry points called by `fn` run directly
        try:
            with torch.no_grad():
                return (impl_fn or fn)(*op_args)
        finally:
            _STATE.inside = False

    _LIB.impl(name, impl, 'CUDA')
    torch.library.register_fake('ever_amd::' + name, fake, lib=_LIB)

    # Autograd for the operator ("op definitions + autograd wrappers"): a direct call of
    # torch.ops.ever_amd.<name> on tensors that require grad differentiates.  The training path of this package does NOT come
    # through here (its autograd.Function wrappers keep the forward's state — statistics records, packed operands, gradient
    # slots — and cost no recomputation); the operator's backward re-runs the entry point under autograd and differentiates
    # that: the same kernels, hence the same gradients bit for bit, at the price of a second forward.
    def setup_context(ctx, inputs, output):
        ctx.spec = [None if isinstance(a, torch.Tensor) else a for a in inputs]
        ctx.is_tensor = [isinstance(a, torch.Tensor) for a in inputs]
        ctx.save_for_backward(*[a for a in inputs if isinstance(a, torch.Tensor)])

    def backward(ctx, grad):
        saved = iter(ctx.saved_tensors)
        args, wrt = [], []
        for i, (is_t, v) in enumerate(zip(ctx.is_tensor, ctx.spec)):
            if is_t:
                t = next(saved).detach()
                if ctx.needs_input_grad[i]:
                    t.requires_grad_()
                    wrt.append((i, t))
                args.append(t)
            else:
                args.append(v)
        _STATE.inside = True
        try:
            with torch.enable_grad():
                y = (impl_fn or fn)(*args)
                gs = torch.autograd.grad(y, [t for _, t in wrt], grad.contiguous(memory_format=torch.channels_last)
                                         if grad.dim() == 4 else grad, allow_unused=True)
        finally:
            _STATE.inside = False
        out = [None] * len(args)
        for (i, _), g in zip(wrt, gs):
            out[i] = g
        return tuple(out)

    torch.library.register_autograd('ever_amd::' + name, backward, setup_context=setup_context, lib=_LIB)
    op = getattr(torch.ops.ever_amd, name)
    OPS[name] = op

    def call(*args, **kwargs):
        if tracing() and not torch.is_grad_enabled() and (applies is None or applies(*args, **kwargs)):
            return op(*adapt(*args, **kwargs))
        return fn(*args, **kwargs)

    call.__name__ = getattr(fn, '__name__', name)
    call.__doc__ = fn.__doc__
    call.__wrapped__ = fn
    return call


def nhwc_like(x, n, c, h, w):
    """fake result: fp32 [n, c, h, w] with channels_last strides on x's device"""
    return x.new_empty((n, c, h, w), dtype=torch.float32).contiguous(memory_format=torch.channels_last)
