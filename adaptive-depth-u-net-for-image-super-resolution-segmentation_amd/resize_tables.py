"""Host-side tap tables for the antialiased bilinear resize kernel (ad_resample).

``tf.image.resize(x, size, method="bilinear", antialias=True)`` -- the op behind ResizeByScale and
ResizeToMatch (/root/reference/shared/custom_layers.py:102,124) -- is TensorFlow's ScaleAndTranslate
with a triangle kernel: per axis a banded [out, in] matrix whose rows are renormalised triangle
weights, with every scalar computed in float32.  This module builds that matrix in banded form
(start index + fixed-width weight rows) and its transpose (the gradient map).
"""
from __future__ import annotations

import functools
import math
from typing import Tuple

import numpy as np


def resized_extent(size: int, scale: float) -> int:
    """ResizeByScale output extent: max(ceil(float32(size) * float32(scale)), 1)  (custom_layers.py:98-101)."""
    return max(int(np.ceil(np.float32(size) * np.float32(scale))), 1)


@functools.lru_cache(maxsize=None)
def aa_spans(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """(starts[out] int32, weights[out, span] float32): y[o] = sum_k weights[o,k] * x[starts[o]+k]."""
    f32 = np.float32
    scale = f32(out_size) / f32(in_size)
    inv_scale = f32(1.0) / scale
    kscale = np.maximum(inv_scale, f32(1.0))  # antialias: widen the triangle only when shrinking
    span = min(2 * int(math.ceil(float(kscale))) + 1, in_size)
    o = np.arange(out_size, dtype=np.float32)
    centre = (o + f32(0.5)) * inv_scale
    lo = np.ceil(centre - kscale - f32(0.5)).astype(np.int64)
    hi = np.floor(centre + kscale - f32(0.5)).astype(np.int64)
    lo = np.clip(lo, 0, in_size - 1)
    hi = np.clip(hi, 0, in_size - 1) + 1
    src = lo[:, None] + np.arange(span, dtype=np.int64)[None, :]
    pos = ((src.astype(np.float32) + f32(0.5)) - centre[:, None]) * (f32(1.0) / kscale)
    wgt = np.maximum(f32(0.0), f32(1.0) - np.abs(pos)).astype(np.float32)
    wgt[src >= hi[:, None]] = 0.0
    total = np.zeros(out_size, dtype=np.float32)
    for k in range(span):  # sequential float32 accumulation, as the TF kernel does
        total = (total + wgt[:, k]).astype(np.float32)
    ok = np.abs(total) >= 1000.0 * np.finfo(np.float32).tiny
    wgt = np.where(ok[:, None], wgt * (f32(1.0) / np.where(ok, total, f32(1.0)))[:, None], f32(0.0)).astype(np.float32)
    outside = (centre < 0) | (centre > in_size)
    wgt[outside] = 0.0
    used = np.flatnonzero((wgt != 0).any(axis=0))          # the nominal span is one tap wider than integer ratios use
    wgt = np.ascontiguousarray(wgt[:, :max(int(used[-1]) + 1 if used.size else 1, 1)])
    return lo.astype(np.int32), wgt


@functools.lru_cache(maxsize=None)
def aa_spans_transposed(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Banded form of the transpose: dx[j] = sum_k wt[j,k] * dy[st[j]+k]  (st[in], wt[in, kmax])."""
    starts, wgt = aa_spans(in_size, out_size)
    dense = np.zeros((out_size, in_size), dtype=np.float32)
    rows = np.repeat(np.arange(out_size), wgt.shape[1])
    cols = (starts[:, None] + np.arange(wgt.shape[1])[None, :]).reshape(-1)
    keep = cols < in_size
    np.add.at(dense, (rows[keep], cols[keep]), wgt.reshape(-1)[keep])
    nz = dense != 0
    first = np.where(nz.any(axis=0), nz.argmax(axis=0), 0)
    last = np.where(nz.any(axis=0), out_size - 1 - nz[::-1].argmax(axis=0), 0)
    kmax = int((last - first + 1).max())
    wt = np.zeros((in_size, kmax), dtype=np.float32)
    for k in range(kmax):
        r = first + k
        valid = r <= last
        wt[valid, k] = dense[np.minimum(r, out_size - 1)[valid], np.arange(in_size)[valid]]
    return first.astype(np.int32), wt


@functools.lru_cache(maxsize=None)
def up_taps2(in_size: int, out_size: int):
    """The up-resize's table in two-tap form for ad_upconv_gather_fwd: (starts[out] int32, weights[out, 2] float32) with
    y[o] = w[o,0] x[starts[o]] + w[o,1] x[min(starts[o] + 1, in - 1)], or None when some output index reads more than two
    consecutive inputs (a shrinking resize).  Leading zero-weight taps of aa_spans are skipped, the weights themselves are
    the float32 values of aa_spans (TensorFlow's ScaleAndTranslate spans, see there)."""
    if out_size < in_size:
        return None
    starts, wgt = aa_spans(in_size, out_size)
    nz = wgt != 0
    first = np.where(nz.any(axis=1), nz.argmax(axis=1), 0)
    rows = np.arange(out_size)
    w2 = np.zeros((out_size, 2), dtype=np.float32)
    w2[:, 0] = wgt[rows, first]
    second = first + 1
    ok2 = second < wgt.shape[1]
    w2[ok2, 1] = wgt[rows[ok2], second[ok2]]
    rest = nz.copy()
    rest[rows, first] = False
    rest[rows[ok2], second[ok2]] = False
    if rest.any():
        return None
    s2 = (starts.astype(np.int64) + first).astype(np.int32)
    w2[(s2 + 1 > in_size - 1), 1] = 0.0      # (no such input: aa_spans never weights it, the kernel clamps the index)
    return s2, w2
