"""Portable deterministic tensors: integer counter-hash -> floats, keyed by a string.

Pure uint64 arithmetic in numpy (splitmix64), so the same name/shape gives bit-identical values on
any machine; weights and inputs of the golden fixtures are regenerated from this instead of being
committed.  TEST INFRASTRUCTURE ONLY.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(key, n, stream=0):
    """n doubles in [0,1) determined by (key, stream, index)."""
    seed = np.uint64((zlib.crc32(key.encode('utf-8')) * 0x100000001 + stream * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over='ignore'):
        idx = (np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + seed) & _M64
    bits = _splitmix64(_splitmix64(idx))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(key, shape, lo=-1.0, hi=1.0, stream=0):
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + (hi - lo) * uniform01(key, n, stream)).astype(np.float32).reshape(shape)


def normalish(key, shape):
    """Irwin-Hall(4) rescaled to unit variance: arithmetic only, bit-portable stand-in for N(0,1)."""
    n = int(np.prod(shape))
    s = sum(uniform01(key, n, stream=k) for k in range(4))
    return ((s - 2.0) * np.sqrt(3.0)).astype(np.float32).reshape(shape)


def sign_vector(key, n):
    """n hash-generated +-1 values (float64): a fixed random direction for gradient projections."""
    return np.where(uniform01(key + '/sign', n) < 0.5, -1.0, 1.0)


def integers(key, shape, hi):
    n = int(np.prod(shape))
    return np.minimum((uniform01(key, n) * hi).astype(np.int64), hi - 1).reshape(shape)


def fill_state_dict(state_dict):
    """Deterministic values for every entry of a FarSeg-style state dict, keyed by its name.
    conv weights: He-uniform on fan_in; BN gamma in [0.5,1.5], beta in [-0.2,0.2]; running stats
    mild; counters zero.  Returns {name: np.ndarray} (same shapes)."""
    out = {}
    for name, t in state_dict.items():
        shape = tuple(t.shape)
        if name.endswith('num_batches_tracked'):
            out[name] = np.zeros(shape, dtype=np.int64)
        elif name.endswith('running_mean'):
            out[name] = uniform(name, shape, -0.1, 0.1)
        elif name.endswith('running_var'):
            out[name] = uniform(name, shape, 0.5, 1.5)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            b = float(np.sqrt(6.0 / fan_in))
            out[name] = uniform(name, shape, -b, b)
        elif name.endswith('weight'):  # BN gamma
            out[name] = uniform(name, shape, 0.5, 1.5)
        else:  # biases
            out[name] = uniform(name, shape, -0.2, 0.2)
    return out


def synthetic_batch(key, n, c, h, w, num_classes=1, p_fg=0.3, ignore_block=8):
    """Image ~N(0,1)-ish [n,c,h,w] f32 and labels [n,h,w] int64 with an ignore(255) corner block
    (SURVEY §8 d2)."""
    x = normalish(key + '/x', (n, c, h, w))
    if num_classes <= 1:
        y = (uniform01(key + '/y', n * h * w) < p_fg).astype(np.int64).reshape(n, h, w)
    else:
        y = integers(key + '/y', (n, h, w), num_classes)
    if ignore_block:
        y[:, :ignore_block, :ignore_block] = 255
    return x, y


# ------------------------------------------------------------------ mask-decision margins of a fixture
def mask_margin(logits):
    """Per-pixel decision margin of a prediction: |logit| for one channel (threshold 0), top-1 minus top-2 otherwise."""
    lg = np.asarray(logits, dtype=np.float64)
    if lg.shape[1] == 1:
        return np.abs(lg[:, 0])
    srt = np.sort(lg, axis=1)
    return srt[:, -1] - srt[:, -2]


def widest_gap_shift(logits, window=0.05):
    """Binary head: the bias shift s with |s| <= window * max|logit| that puts the threshold in the middle of the WIDEST
    empty interval of the logit values (so that no pixel of the fixture sits near the decision).  A shift of the 1x1
    classifier's bias moves every logit by s (bilinear weights sum to one).  Returns (s, half-width of the gap)."""
    v = np.sort(np.asarray(logits, dtype=np.float64).reshape(-1))
    r = np.abs(v).max()
    lo, hi = np.searchsorted(v, -window * r), np.searchsorted(v, window * r)
    seg = v[max(lo - 1, 0):hi + 1]
    gaps = np.diff(seg)
    i = int(np.argmax(gaps))
    mid = 0.5 * (seg[i] + seg[i + 1])
    return float(-mid), float(0.5 * gaps[i])


def widest_gap_bias(logits, key, tries=400, scale=0.03):
    """Multi-class head: among `tries` hash-generated per-class bias vectors (|b| <= scale * logit range) the one that
    maximises the smallest top-1 / top-2 gap over the fixture's pixels.  Returns (bias [C] float32, smallest gap)."""
    lg = np.asarray(logits, dtype=np.float64)
    c = lg.shape[1]
    r = np.abs(lg).max()
    best, best_gap = np.zeros(c, dtype=np.float32), float(mask_margin(lg).min())
    for t in range(tries):
        b = uniform(f'{key}/bias{t}', (c,), -scale * r, scale * r)
        g = float(mask_margin(lg + b.astype(np.float64).reshape(1, c, 1, 1)).min())
        if g > best_gap:
            best, best_gap = b, g
    return best, best_gap
