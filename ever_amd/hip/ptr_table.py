"""Address / size tables for the multi-tensor kernels (evk_sgd_multi, evk_sqnorm_multi, evk_pack_multi)."""
import torch

__all__ = ['PtrTable']


class PtrTable:
    """int64 table in pinned host memory + its device copy; `upload` re-sends only when the values changed."""

    def __init__(self, n, dev):
        self.n, self.dev = n, dev
        self.host = torch.empty((max(n, 1),), dtype=torch.int64, pin_memory=True)
        self.device = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
        self.last = None
        self.event = None

    def upload(self, vals):
        if vals != self.last:
            if self.event is not None:
                self.event.synchronize()  # the previous upload has left the pinned buffer (long done in practice)
            self.host[:self.n] = torch.tensor(vals, dtype=torch.int64)
            self.device.copy_(self.host, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
            self.last = list(vals)
        return self.device
