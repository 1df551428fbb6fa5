"""HIP (gfx950) execution layer: ctypes-bound C-ABI kernels wrapped as autograd functions."""
from . import functional  # noqa: F401
from .functional import HipPathError  # noqa: F401
