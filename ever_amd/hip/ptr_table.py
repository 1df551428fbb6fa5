"""Address / size tables for the multi-tensor kernels (evk_sgd_multi, evk_sqnorm_multi, evk_pack_multi)."""
import torch

__all__ = ['PtrTable']


class PtrTable:
    """int64 table in pinned host memory + its device copy; `upload` re-sends only when the values changed.

    Under hipGraph capture (core/graph.py) the upload becomes a memcpy node that re-reads the pinned buffer at every replay:
    it goes through a SECOND pinned buffer (allocated here, up front: pinned allocation is not permitted while a stream
    captures) that eager uploads never touch, and no event is recorded or waited for inside the capture."""

    def __init__(self, n, dev):
        self.n, self.dev = n, dev
        self.host = torch.empty((max(n, 1),), dtype=torch.int64, pin_memory=True)
        self.host_graph = torch.empty((max(n, 1),), dtype=torch.int64, pin_memory=True)
        self.device = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
        self.last = None
        self.event = None

    def upload(self, vals):
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            # always part of the graph: the device table must hold these values at every replay, whatever an eager call
            # in between has put there
            self.host_graph[:self.n] = torch.tensor(vals, dtype=torch.int64)
            self.device.copy_(self.host_graph, non_blocking=True)
            self.last = None
            return self.device
        if vals != self.last:
            if self.event is not None:
                self.event.synchronize()  # the previous upload has left the pinned buffer (long done in practice)
            self.host[:self.n] = torch.tensor(vals, dtype=torch.int64)
            self.device.copy_(self.host, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
            self.last = list(vals)
        return self.device
