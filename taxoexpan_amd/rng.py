"""Host restatement of the counter-based dropout hash of csrc/txe_common.h (mix64 / uniform01 / drop_factor).

The kernels never store a dropout mask: keep(seed, index) is a pure function that forward and backward both
re-evaluate.  This numpy version produces the identical mask so that tests can hand the very same mask to the
oracle (whose dropout is an explicit keep-mask multiply, model_zoo.py:82,114).
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return z ^ (z >> np.uint64(31))


def uniform01(seed, idx):
    """idx: integer array.  Returns float32 uniforms in [0,1) exactly as the device computes them."""
    with np.errstate(over="ignore"):
        idx = np.asarray(idx).astype(np.uint64)
        h = _mix64(np.uint64(seed) ^ _mix64(idx))
    return (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def keep_mask(seed, shape, p):
    """0/1 float32 keep mask of `shape` (row-major linear index), for dropout probability p."""
    n = int(np.prod(shape))
    u = uniform01(seed, np.arange(n, dtype=np.uint64))
    return (u >= np.float32(p)).astype(np.float32).reshape(shape)


def keep_mask_bits(seed, n_rows, n_cols, p):
    """The feature-dropout keep mask of txe_dropout_mask (csrc/txe_common.h drop_mask_word) as a 0/1 float32
    [n_rows, n_cols] array.  Bit b of mask word w is [u(w, b) >= round(p*65536)] for a 16-bit uniform u(w, b) whose bit j is bit b of
    the plane word R(w, j) = 32-bit half (j & 1) of mix64(seed + (8w + (j >> 1)) * W); evaluated plane-wise from the threshold's
    lowest set bit upwards: ge = ge & R_j where the threshold has a 1, ge | R_j where it has a 0."""
    wpr = (n_cols + 31) // 32
    n_words = n_rows * wpr
    thr = int(np.float32(p) * np.float32(65536.0) + np.float32(0.5))
    if thr == 0:
        ge = np.full(n_words, 0xFFFFFFFF, dtype=np.uint64)
    elif thr > 0xFFFF:
        ge = np.zeros(n_words, dtype=np.uint64)
    else:
        j0 = (thr & -thr).bit_length() - 1
        ge = np.full(n_words, 0xFFFFFFFF, dtype=np.uint64)
        w = np.arange(n_words, dtype=np.uint64)
        with np.errstate(over="ignore"):
            for k in range(j0 >> 1, 8):
                h = _mix64((np.uint64(seed) + (w * np.uint64(8) + np.uint64(k)) * np.uint64(0xD1342543DE82EF95)) & _M)
                for half in range(2):
                    j = 2 * k + half
                    if j < j0:
                        continue
                    r = (h >> np.uint64(32 * half)) & np.uint64(0xFFFFFFFF)
                    ge = (ge & r) if (thr >> j) & 1 else (ge | r)
    bits = ((ge[:, None] >> np.arange(32, dtype=np.uint64)[None, :]) & np.uint64(1)).reshape(n_rows, wpr * 32)
    return bits[:, :n_cols].astype(np.float32)
