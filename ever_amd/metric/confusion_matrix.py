"""Confusion matrix accumulated on the GPU (API of reference ever/metric/confusion_matrix.py:6-39).

The reference moves every batch to the host and adds a scipy.sparse COO matrix per call; here the counts
are added into an int64 [C, C] device tensor by evk_confusion_matrix / evk_confusion_from_logits (exact
integer arithmetic, no host sync per batch) and only `dense_cm` copies them out.  CPU inputs (numpy or CPU
tensors) are counted on the host with numpy, as the reference does — evaluation utilities are host code."""
import numpy as np
import torch

from ..hip import functional as HF

__all__ = ['ConfusionMatrix']


class ConfusionMatrix(object):
    def __init__(self, num_classes, device=None):
        self.num_classes = num_classes
        self._device = torch.device(device) if device is not None else None
        self._gpu_total = None                                   # int64 [C, C] on the GPU, created lazily
        self._host_total = np.zeros((num_classes, num_classes), dtype=np.int64)

    def _gpu(self, device):
        if self._gpu_total is None:
            self._gpu_total = torch.zeros((self.num_classes, self.num_classes), dtype=torch.int64, device=device)
        return self._gpu_total

    def forward(self, y_true, y_pred):
        """Accumulate one batch; row = ground truth, column = prediction.  Returns the batch's own matrix
        as a dense float32 array (the reference returns the sparse batch matrix)."""
        c = self.num_classes
        if isinstance(y_pred, torch.Tensor) and y_pred.is_cuda:
            batch = torch.zeros((c, c), dtype=torch.int64, device=y_pred.device)
            HF.confusion_matrix_update(batch, y_true, y_pred=y_pred)
            self._gpu(y_pred.device).add_(batch)
            return _LazyDense(batch)
        yt = y_true.cpu().numpy() if isinstance(y_true, torch.Tensor) else np.asarray(y_true)
        yp = y_pred.cpu().numpy() if isinstance(y_pred, torch.Tensor) else np.asarray(y_pred)
        yt, yp = yt.reshape(-1).astype(np.int64), yp.reshape(-1).astype(np.int64)
        ok = (yt >= 0) & (yt < c) & (yp >= 0) & (yp < c)
        batch = np.bincount(yt[ok] * c + yp[ok], minlength=c * c).reshape(c, c).astype(np.int64)
        self._host_total += batch
        return batch.astype(np.float32)

    def forward_logits(self, y_true, logits):
        """Fused evaluation step: threshold-0 / argmax prediction and counting in one kernel over NCHW logits."""
        HF.confusion_matrix_update(self._gpu(logits.device), y_true, logits=logits)

    @property
    def dense_cm(self):
        total = self._host_total.copy()
        if self._gpu_total is not None:
            total += self._gpu_total.cpu().numpy()
        return total.astype(np.float32)

    @property
    def sparse_cm(self):
        from scipy import sparse
        return sparse.coo_matrix(self.dense_cm)

    def reset(self):
        self._host_total[...] = 0
        if self._gpu_total is not None:
            self._gpu_total.zero_()

    @staticmethod
    def plot(confusion_matrix):
        return NotImplementedError


class _LazyDense(object):
    """Batch matrix that stays on the GPU until somebody looks at it."""

    def __init__(self, t):
        self._t = t

    def toarray(self):
        return self._t.cpu().numpy().astype(np.float32)

    def __array__(self, dtype=None):
        a = self.toarray()
        return a.astype(dtype) if dtype is not None else a
