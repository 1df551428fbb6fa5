"""Tiling of a large image into fixed-size windows (reference ever/magic/bigimage/sliding_window.py:8-33) and
a tiled-inference driver on top of it (the consumer of a trained checkpoint)."""
import math

import numpy as np
import torch


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _starts(extent, k, s):
    """Window origins along one axis: a regular grid with stride s, windows that would overrun the image are
    pulled back so they end at the border (so the last two windows may overlap more, or coincide)."""
    steps = math.ceil((extent - k) / s)
    n = steps if steps * s + k >= extent else steps + 1
    origins = np.arange(n + 1) * s
    return np.where(origins + k > extent, extent - k, origins), np.minimum(np.arange(n + 1) * s + k, extent)


def sliding_window(input_size, kernel_size, stride):
    """boxes [M, 4] = (xmin, ymin, xmax, ymax), row-major over the window grid; a kernel larger than the image
    is clipped to it.  Same enumeration (including the duplicated border windows) as the reference."""
    ih, iw = input_size
    kh, kw = _pair(kernel_size)
    sh, sw = _pair(stride)
    assert ih > 0 and iw > 0 and kh > 0 and kw > 0 and sh > 0 and sw > 0
    kh, kw = min(kh, ih), min(kw, iw)
    y0, y1 = _starts(ih, kh, sh)
    x0, x1 = _starts(iw, kw, sw)
    gx0, gy0 = np.meshgrid(x0, y0)
    gx1, gy1 = np.meshgrid(x1, y1)
    return np.stack([gx0.ravel(), gy0.ravel(), gx1.ravel(), gy1.ravel()], axis=1)


@torch.no_grad()
def sliding_window_inference(model, image, kernel_size, stride, batch_size=8, activation=None):
    """Run `model` (eval mode, returns [B, C, h, w] scores) over the windows of `image` [1|N, C, H, W] and
    average the overlapping scores.  Windows are batched; accumulation happens on the image's device."""
    assert image.dim() == 4
    n, _, ih, iw = image.shape
    boxes = np.unique(sliding_window((ih, iw), kernel_size, stride), axis=0)
    out = count = None
    for b in range(n):
        for i in range(0, len(boxes), batch_size):
            chunk = boxes[i:i + batch_size]
            tiles = torch.stack([image[b, :, y0:y1, x0:x1] for x0, y0, x1, y1 in chunk])
            scores = model(tiles)
            if activation is not None:
                scores = activation(scores)
            if out is None:
                out = torch.zeros((n, scores.shape[1], ih, iw), device=scores.device, dtype=scores.dtype)
                count = torch.zeros((1, 1, ih, iw), device=scores.device, dtype=scores.dtype)
            for s, (x0, y0, x1, y1) in zip(scores, chunk):
                out[b, :, y0:y1, x0:x1] += s
                if b == 0:
                    count[0, 0, y0:y1, x0:x1] += 1
    return out / count
