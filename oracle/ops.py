"""NumPy restatement of every tensor op on the hot path (oracle; test infrastructure only).

All tensors are NHWC.  Every function works in the dtype of its inputs (tests use
float64 for the ground truth and float32 to mimic the reference's CPU path).
Each function cites the reference call site (paths relative to /root/reference)
whose TensorFlow/Keras op it restates.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

# --------------------------------------------------------------------------- #
# Storage rounding of the reduced-precision product paths.  The reference's mixed-precision policy
# (Super_resolution/code/train_adaptive_unet.py:471-477) keeps variables in float32 and STORES activations in a
# 16-bit type; the arithmetic inside an op is wider.  The oracle mimics that by rounding tensors at exactly the
# points where the product stores them, while every sum stays float64.
# --------------------------------------------------------------------------- #


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round to the nearest bfloat16 (ties to even), returned in x's dtype."""
    if isinstance(x, np.ndarray) and x.dtype == np.float64 and x.size >= (1 << 20) and _conv_c():
        xc = np.ascontiguousarray(x)                      # large float64 tensors: the same arithmetic in one parallel C pass
        out = np.empty_like(xc)
        _conv_c().oracle_bf16_round(xc.ctypes.data, out.ctypes.data, xc.size)
        return out
    f = np.ascontiguousarray(x, dtype=np.float32)
    u = f.view(np.uint32)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    out = r.view(np.float32)
    out = np.where(np.isfinite(f), out, f)
    return out.astype(x.dtype) if hasattr(x, "dtype") else out


def fp16_round(x: np.ndarray) -> np.ndarray:
    """Round to the nearest IEEE half (ties to even; overflow -> inf, as the hardware conversion does)."""
    with np.errstate(over="ignore"):
        return np.asarray(x).astype(np.float16).astype(np.asarray(x).dtype)


# --------------------------------------------------------------------------- #
# Conv2D 3x3 / 1x1, stride 1, padding "same", bias
#   Super_resolution/code/train_adaptive_unet.py:202,207,259,267-274
#   kernel layout HWIO (Keras), zero padding.
# --------------------------------------------------------------------------- #


_CONV_C = None


def cpu_share(cap: int = 16) -> int:
    """Threads worth starting in this process: the cgroup's CPU quota where there is one (the GPU boxes show 256 logical CPUs to a
    process whose share is 16), else the visible CPUs, never more than `cap` (ORACLE_THREADS overrides)."""
    import os
    if os.environ.get("ORACLE_THREADS"):
        return max(1, int(os.environ["ORACLE_THREADS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, min(n, cap))


def _conv_c():
    """The C restatement of the same sums (oracle/csrc/conv_ref.c, built by oracle/Makefile / __graft_entry__.build()), used for
    float64 tensors large enough to matter: the layer-wise audits convolve whole BASELINE-size batches, and NumPy's strided tap
    copies made that most of the GPU suite's run time.  False when the library has not been built (NumPy path below)."""
    global _CONV_C
    if _CONV_C is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_c", "liboracle_conv.so")
        _CONV_C = False
        if os.path.exists(path) and os.environ.get("ORACLE_NO_C") != "1":
            lib = ctypes.CDLL(path)
            vp, lg = ctypes.c_void_p, ctypes.c_long
            lib.oracle_conv_fwd.argtypes = [vp, vp, vp, vp, lg, lg, lg, lg, lg, lg, lg]
            lib.oracle_conv_wgrad.argtypes = [vp, vp, vp, lg, lg, lg, lg, lg, lg, lg]
            lib.oracle_conv_fwd.restype = lib.oracle_conv_wgrad.restype = ctypes.c_int
            lib.oracle_ln_fwd.argtypes = [vp, vp, vp, ctypes.c_double, vp, vp, vp, lg, lg]
            lib.oracle_ln_fwd.restype = None
            lib.oracle_ln_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, lg, lg]
            lib.oracle_ln_bwd.restype = ctypes.c_int
            lib.oracle_bf16_round.argtypes = [vp, vp, lg]
            lib.oracle_bf16_round.restype = None
            lib.oracle_set_threads.argtypes = [ctypes.c_int]
            lib.oracle_set_threads.restype = None
            lib.oracle_set_threads(cpu_share())
            _CONV_C = lib
    return _CONV_C


def _use_c(x: np.ndarray, w: np.ndarray) -> bool:
    # float64 only; the per-thread private dw copies of the C wgrad stay small (<= 8 MB each), and below ~0.2 GFLOP NumPy is fine
    kh, kw, cin, cout = w.shape
    return (x.dtype == np.float64 and w.dtype == np.float64 and kh * kw * cin * cout <= (1 << 20)
            and x.size // cin * kh * kw * cin * cout >= (1 << 27) and bool(_conv_c()))


def _ptr(a: np.ndarray):
    return a.ctypes.data


def conv2d_same_fwd(x: np.ndarray, w: np.ndarray, b: np.ndarray | None) -> np.ndarray:
    """One [pixels, Cin] x [Cin, Cout] product per tap (the shifted view is copied to a dense matrix first, so that the
    product is a single multi-threaded GEMM); large float64 cases run the same sums in C (_conv_c)."""
    kh, kw, cin, cout = w.shape
    n, h, wd, c = x.shape
    assert c == cin, (x.shape, w.shape)
    ph, pw = kh // 2, kw // 2
    if _use_c(x, w):
        y = np.empty((n, h, wd, cout), dtype=np.float64)
        xc, wc = np.ascontiguousarray(x), np.ascontiguousarray(w)
        bc = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
        rc = _conv_c().oracle_conv_fwd(_ptr(xc), _ptr(wc), _ptr(bc) if bc is not None else None, _ptr(y), n, h, wd, cin, cout, kh, kw)
        assert rc == 0, "oracle_conv_fwd: out of memory"
        return y
    xp = np.pad(x, ((0, 0), (ph, ph), (pw, pw), (0, 0)))
    y = np.zeros((n * h * wd, cout), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            y += np.ascontiguousarray(xp[:, i:i + h, j:j + wd, :]).reshape(-1, cin) @ w[i, j]
    if b is not None:
        y += b
    return y.reshape(n, h, wd, cout)


def conv2d_same_bwd(x: np.ndarray, w: np.ndarray, dy: np.ndarray, need_dx: bool = True):
    """Returns (dx, dw, db) -- dgrad, wgrad and bias grad of conv2d_same_fwd."""
    kh, kw, cin, cout = w.shape
    n, h, wd, _ = x.shape
    ph, pw = kh // 2, kw // 2
    if _use_c(x, w) and dy.dtype == np.float64:
        dyc, xc = np.ascontiguousarray(dy), np.ascontiguousarray(x)
        dw = np.empty(w.shape, dtype=np.float64)
        rc = _conv_c().oracle_conv_wgrad(_ptr(xc), _ptr(dyc), _ptr(dw), n, h, wd, cin, cout, kh, kw)
        assert rc == 0, "oracle_conv_wgrad: out of memory"
        dx = None
        if need_dx:      # dgrad = the forward sum over dy with the kernel rotated by 180 degrees and its channel axes swapped
            dx = conv2d_same_fwd(dyc, np.ascontiguousarray(w[::-1, ::-1].transpose(0, 1, 3, 2)), None)
        return dx, dw, dyc.reshape(-1, cout).sum(axis=0)
    xp = np.pad(x, ((0, 0), (ph, ph), (pw, pw), (0, 0)))
    dw = np.zeros_like(w)
    dxp = np.zeros_like(xp) if need_dx else None
    dy2 = np.ascontiguousarray(dy).reshape(-1, cout)
    for i in range(kh):
        for j in range(kw):
            patch = np.ascontiguousarray(xp[:, i:i + h, j:j + wd, :]).reshape(-1, cin)
            dw[i, j] = patch.T @ dy2
            if need_dx:
                dxp[:, i:i + h, j:j + wd, :] += (dy2 @ w[i, j].T).reshape(n, h, wd, cin)
    db = dy2.sum(axis=0)
    dx = dxp[:, ph:ph + h, pw:pw + wd, :] if need_dx else None
    return dx, dw, db


# --------------------------------------------------------------------------- #
# LayerNormalization(axis=-1), Keras default epsilon 1e-3, biased variance
#   Super_resolution/code/train_adaptive_unet.py:203,208
# --------------------------------------------------------------------------- #

LN_EPS = 1e-3


def _ln_use_c(x, *vecs) -> bool:
    # float64 tensors of a million elements or more (the whole-batch tensors of the layer-wise audits): one fused C pass per
    # pixel (oracle/csrc/conv_ref.c) instead of a dozen whole-tensor temporaries; everything else runs the NumPy lines below
    return (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.size >= (1 << 20) and x.ndim >= 2
            and all(np.asarray(v).shape == (x.shape[-1],) for v in vecs) and bool(_conv_c()))


def layernorm_fwd(x, gamma, beta, eps: float = LN_EPS):
    if _ln_use_c(x, gamma, beta):
        xc = np.ascontiguousarray(x)
        g, b = np.ascontiguousarray(gamma, dtype=np.float64), np.ascontiguousarray(beta, dtype=np.float64)
        y, xhat = np.empty_like(xc), np.empty_like(xc)
        rstd = np.empty(xc.shape[:-1] + (1,), dtype=np.float64)
        _conv_c().oracle_ln_fwd(_ptr(xc), _ptr(g), _ptr(b), float(eps), _ptr(y), _ptr(xhat), _ptr(rstd), xc.size // xc.shape[-1],
                                xc.shape[-1])
        return y, (xhat, rstd)
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mu) * rstd
    return xhat * gamma + beta, (xhat, rstd)


def layernorm_bwd(dy, gamma, cache):
    xhat, rstd = cache
    if (_ln_use_c(dy, gamma) and isinstance(xhat, np.ndarray) and xhat.dtype == np.float64 and xhat.shape == dy.shape
            and isinstance(rstd, np.ndarray) and rstd.dtype == np.float64 and rstd.size == dy.size // dy.shape[-1]):
        dyc, hc, rc = np.ascontiguousarray(dy), np.ascontiguousarray(xhat), np.ascontiguousarray(rstd)
        g = np.ascontiguousarray(gamma, dtype=np.float64)
        c = dyc.shape[-1]
        dx, dgamma, dbeta = np.empty_like(dyc), np.empty(c), np.empty(c)
        rc_ = _conv_c().oracle_ln_bwd(_ptr(dyc), _ptr(g), _ptr(hc), _ptr(rc), _ptr(dx), _ptr(dgamma), _ptr(dbeta), dyc.size // c, c)
        assert rc_ == 0, "oracle_ln_bwd: out of memory"
        return dx, dgamma, dbeta
    red = tuple(range(dy.ndim - 1))
    dgamma = (dy * xhat).sum(axis=red)
    dbeta = dy.sum(axis=red)
    g = dy * gamma
    dx = rstd * (g - g.mean(axis=-1, keepdims=True) - xhat * (g * xhat).mean(axis=-1, keepdims=True))
    return dx, dgamma, dbeta


def relu_fwd(x):
    return np.maximum(x, 0)


def relu_bwd(dy, y):
    # TF ReluGrad: gradient passes where the feature is strictly positive.
    return dy * (y > 0)


# --------------------------------------------------------------------------- #
# tf.image.resize(..., method="bilinear", antialias=True)
#   shared/custom_layers.py:93-103 (ResizeByScale), :121-125 (ResizeToMatch)
#   = gen_image_ops.scale_and_translate(kernel_type="triangle", antialias=True)
#   Restated from TF's published ScaleAndTranslate kernel (ComputeSpansCore):
#   all span/weight arithmetic is float32; weights renormalised per output index.
# --------------------------------------------------------------------------- #


def resize_by_scale_size(h: int, scale: float) -> int:
    """shared/custom_layers.py:98-101 -- ceil in float32, at least 1."""
    v = np.float32(h) * np.float32(scale)
    return max(int(np.ceil(v)), 1)


def aa_triangle_spans(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Per-output span starts and normalised float32 weights.

    Returns (starts[out], weights[out, span]) with zero padding after each span.
    """
    f32 = np.float32
    scale = f32(out_size) / f32(in_size)
    inv_scale = f32(1.0) / scale
    kernel_scale = max(inv_scale, f32(1.0))  # antialias=True
    radius = f32(1.0)
    span = min(2 * int(math.ceil(float(radius * kernel_scale))) + 1, in_size)
    starts = np.zeros(out_size, dtype=np.int64)
    weights = np.zeros((out_size, span), dtype=np.float32)
    one_over = f32(1.0) / kernel_scale
    for o in range(out_size):
        sample = (f32(o) + f32(0.5)) * inv_scale
        if sample < 0 or sample > in_size:
            continue
        s0 = int(math.ceil(float(sample - radius * kernel_scale - f32(0.5))))
        s1 = int(math.floor(float(sample + radius * kernel_scale - f32(0.5))))
        s0 = min(max(s0, 0), in_size - 1)
        s1 = min(max(s1, 0), in_size - 1) + 1
        ws = []
        total = f32(0.0)
        for src in range(s0, s1):
            pos = (f32(src) + f32(0.5) - sample) * one_over
            wgt = max(f32(0.0), f32(1.0) - abs(pos))
            ws.append(wgt)
            total = f32(total + wgt)
        if abs(total) >= 1000.0 * np.finfo(np.float32).tiny:
            for k, wgt in enumerate(ws):
                weights[o, k] = f32(wgt * (f32(1.0) / total))
        starts[o] = s0
    return starts, weights


def aa_matrix(in_size: int, out_size: int, dtype=np.float64) -> np.ndarray:
    """Dense [out, in] interpolation matrix of the 1-D antialiased triangle resize."""
    starts, weights = aa_triangle_spans(in_size, out_size)
    m = np.zeros((out_size, in_size), dtype=dtype)
    for o in range(out_size):
        for k in range(weights.shape[1]):
            j = starts[o] + k
            if j < in_size and weights[o, k] != 0:
                m[o, j] += weights[o, k]
    return m


def _apply_axis(m: np.ndarray, x: np.ndarray, axis: int) -> np.ndarray:
    """y = m applied along `axis` of x (a GEMM on the axis moved to the front)."""
    xm = np.moveaxis(x, axis, 0)
    y = m @ np.ascontiguousarray(xm).reshape(xm.shape[0], -1)
    return np.moveaxis(y.reshape((m.shape[0],) + xm.shape[1:]), 0, axis)


def resize_aa_fwd(x: np.ndarray, oh: int, ow: int) -> np.ndarray:
    n, h, w, c = x.shape
    my = aa_matrix(h, oh, x.dtype)
    mx = aa_matrix(w, ow, x.dtype)
    return np.ascontiguousarray(_apply_axis(mx, _apply_axis(my, x, 1), 2))      # separable: H, then W


def resize_aa_bwd(dy: np.ndarray, h: int, w: int) -> np.ndarray:
    n, oh, ow, c = dy.shape
    my = aa_matrix(h, oh, dy.dtype)
    mx = aa_matrix(w, ow, dy.dtype)
    return np.ascontiguousarray(_apply_axis(my.T, _apply_axis(mx.T, dy, 2), 1))  # the transposed map


# --------------------------------------------------------------------------- #
# Decoder step "up-resize -> Conv3x3 + ReLU" without the up-resized tensor
#   Super_resolution/code/train_adaptive_unet.py:258-259  (dec_up -> L.Conv2D(nf, 3, padding="same", activation="relu"))
#   shared/custom_layers.py:121-125                       (ResizeToMatch)
# The resize acts on pixels, the convolution's contraction on channels, so the two commute:
#   conv3x3(U x)[p] = b + sum_tap (U x)[p + tap] W_tap = b + sum_tap (U (x W_tap))[p + tap]        (zero outside the image)
# i.e. a bank of nine 1x1 convolutions on the LOW-resolution map (Y_tap = x W_tap, 1 / ratio^2 of the conv's FLOPs)
# followed by a gather that interpolates and shifts.  The product computes the step in this form; these functions
# restate it tap by tap (no fused shortcuts) and tests/test_oracle_factored_upconv.py checks them against
# relu(conv2d_same_fwd(resize_aa_fwd(x), w, b)) and its gradients to float64 rounding.
# --------------------------------------------------------------------------- #


def upconv_bank_fwd(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """Y[n, h, w, tap, co] = sum_ci x[n, h, w, ci] W[tap // 3, tap % 3, ci, co]: nine 1x1 convolutions, one GEMM."""
    kh, kw, cin, cout = w.shape
    n, h, wd, _ = x.shape
    bank = np.transpose(w.reshape(kh * kw, cin, cout), (1, 0, 2)).reshape(cin, kh * kw * cout)
    return (np.ascontiguousarray(x).reshape(-1, cin) @ bank).reshape(n, h, wd, kh * kw, cout)


def _shift_rows(m: np.ndarray, d: int) -> np.ndarray:
    """Rows of the [out, in] resize matrix moved so that row p holds the weights of output index p + d (zero rows where
    p + d falls outside the image: the convolution's zero padding lives at the HIGH resolution)."""
    out = np.zeros_like(m)
    n = m.shape[0]
    lo, hi = max(0, -d), min(n, n - d)
    out[lo:hi] = m[lo + d:hi + d]
    return out


def upconv_gather_fwd(y: np.ndarray, b: np.ndarray | None, oh: int, ow: int) -> np.ndarray:
    """out[p] = b + sum_tap (U Y_tap)[p + tap]  (pre-activation)."""
    n, h, w, taps, cout = y.shape
    my, mx = aa_matrix(h, oh, y.dtype), aa_matrix(w, ow, y.dtype)
    out = np.zeros((n, oh, ow, cout), dtype=y.dtype)
    for t in range(taps):
        dy, dx = t // 3 - 1, t % 3 - 1
        out += _apply_axis(_shift_rows(mx, dx), _apply_axis(_shift_rows(my, dy), y[:, :, :, t, :], 1), 2)
    if b is not None:
        out += b
    return out


def upconv_gather_bwd(g: np.ndarray, h: int, w: int) -> np.ndarray:
    """dY[n, h, w, tap, co] from the gradient g of the pre-activation: the transpose of upconv_gather_fwd."""
    n, oh, ow, cout = g.shape
    my, mx = aa_matrix(h, oh, g.dtype), aa_matrix(w, ow, g.dtype)
    dy_ = np.zeros((n, h, w, 9, cout), dtype=g.dtype)
    for t in range(9):
        dy, dx = t // 3 - 1, t % 3 - 1
        dy_[:, :, :, t, :] = _apply_axis(_shift_rows(my, dy).T, _apply_axis(_shift_rows(mx, dx).T, g, 2), 1)
    return dy_


def upconv_bank_bwd(x: np.ndarray, w: np.ndarray, dyb: np.ndarray):
    """(dx, dw) of upconv_bank_fwd: dx = dY W_bank^T, dW_tap = x^T dY_tap."""
    kh, kw, cin, cout = w.shape
    n, h, wd, _ = x.shape
    bank = np.transpose(w.reshape(kh * kw, cin, cout), (1, 0, 2)).reshape(cin, kh * kw * cout)
    d2 = np.ascontiguousarray(dyb).reshape(-1, kh * kw * cout)
    dx = (d2 @ bank.T).reshape(n, h, wd, cin)
    dbank = np.ascontiguousarray(x).reshape(-1, cin).T @ d2
    dw = np.transpose(dbank.reshape(cin, kh * kw, cout), (1, 0, 2)).reshape(kh, kw, cin, cout)
    return dx, dw


# --------------------------------------------------------------------------- #
# ClippedResidualAdd  (shared/custom_layers.py:136-139)
# --------------------------------------------------------------------------- #


def clip_add_fwd(inp, res):
    pre = inp + res
    return np.clip(pre, 0.0, 1.0), pre


def clip_add_bwd(dout, pre):
    # tf.clip_by_value gradient: zero where pre < 0 or pre > 1 (bounds inclusive pass).
    return dout * ((pre >= 0.0) & (pre <= 1.0))


# --------------------------------------------------------------------------- #
# Losses / metrics  (Super_resolution/code/train_adaptive_unet.py:308-334)
# --------------------------------------------------------------------------- #

CHARBONNIER_EPS = 1e-3


def charbonnier_fwd(y_true, y_pred, eps: float = CHARBONNIER_EPS):
    d = y_true - y_pred
    return np.sqrt(d * d + eps * eps).mean()


def charbonnier_bwd(y_true, y_pred, eps: float = CHARBONNIER_EPS):
    d = y_true - y_pred
    return -d / np.sqrt(d * d + eps * eps) / d.size


def l1_fwd(y_true, y_pred):
    return np.abs(y_true - y_pred).mean()


def l1_bwd(y_true, y_pred):
    return -np.sign(y_true - y_pred) / y_true.size


def psnr_per_image(y_true, y_pred, max_val: float = 1.0):
    """tf.image.psnr on clip(y_pred): per-image, inf when MSE == 0 (:308-311).  Float64 truth; the float32 arithmetic
    TensorFlow itself runs (pinned by the reference's evaluation CSVs) is oracle.metrics.psnr_from_mse."""
    yp = np.clip(y_pred, 0.0, 1.0)
    mse = ((y_true - yp) ** 2).reshape(y_true.shape[0], -1).mean(axis=1)
    with np.errstate(divide="ignore"):
        return 20.0 * np.log10(max_val) - 10.0 * np.log10(mse)


def rgb_to_luma_bt601(image):
    """Super_resolution/code/train_adaptive_unet.py:144-157."""
    coeffs = np.array([65.481, 128.553, 24.966], dtype=image.dtype)
    y = (image * coeffs).sum(axis=-1, keepdims=True) + 16.0
    return np.clip(y / 255.0, 0.0, 1.0)


def infer_eval_shave(scale: float) -> int:
    """Super_resolution/code/evaluate_model.py:49-54 (2 * round(1/scale))."""
    if scale <= 0:
        return 0
    return max(0, 2 * int(round(1.0 / scale)))


# --------------------------------------------------------------------------- #
# Keras-form Adam (keras 3.3.3 optimizers/adam.py, via train_adaptive_unet.py:489-494)
# --------------------------------------------------------------------------- #


def adam_step(p, g, m, v, step: int, lr=1e-4, b1=0.9, b2=0.999, eps=1e-7):
    """In-place update; `step` is 1-based (iterations + 1)."""
    alpha = lr * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    m += (g - m) * (1.0 - b1)
    v += (g * g - v) * (1.0 - b2)
    p -= m * alpha / (np.sqrt(v) + eps)


# --------------------------------------------------------------------------- #
# Depth heuristics (shared/custom_layers.py:10-82), pure Python
# --------------------------------------------------------------------------- #


def infer_depth_from_scale(scale: float, min_depth: int = 1, max_depth: int = 4) -> int:
    if not (0.05 < scale < 1.0):
        raise ValueError("Scale should be between 0 and 1 (exclusive).")
    depth = 1 if scale <= 0.25 else (2 if scale <= 0.45 else 3)
    return max(min_depth, min(depth, max_depth))


def depth_and_sizes(scale, min_res=21, max_depth=7):
    depth, sizes, res = 1, [256], 256
    while res > min_res and depth < max_depth:
        res = math.ceil(res * scale)
        sizes.append(res)
        depth += 1
    return min(depth, max_depth), sizes


def custom_depth_from_scale(scale, min_depth=1, max_depth=7, *, base_resolution=256, min_feature=21) -> int:
    if not (0.05 < scale < 1.0):
        raise ValueError("Scale should be between 0 and 1 (exclusive).")
    if min_depth < 1 or max_depth < 1 or base_resolution <= 0 or min_feature < 1:
        raise ValueError("invalid argument")
    depth, extent = max(min_depth, 1), base_resolution
    while depth < max_depth:
        cand = math.ceil(extent * scale)
        if cand < min_feature:
            break
        extent = cand
        depth += 1
    return max(min_depth, min(depth, max_depth))


def estimate_bottleneck_size(hr: int, scale: float, depth: int) -> int:
    size = hr
    for _ in range(depth):
        size = max(1, int(round(size * scale)))
    return size


# --------------------------------------------------------------------------- #
# Tier-2 ops (segmentation models)
#   Segmenation/code/train_adaptive_unet.py:258-362, Segmenation/code/unet_vinillia.py:42-99
# --------------------------------------------------------------------------- #

BN_EPS = 1e-3
BN_MOMENTUM = 0.99


def batchnorm_train_fwd(x, gamma, beta, eps: float = BN_EPS):
    red = (0, 1, 2)
    mu = x.mean(axis=red)
    var = ((x - mu) ** 2).mean(axis=red)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mu) * rstd
    return xhat * gamma + beta, (xhat, rstd), mu, var


def batchnorm_train_bwd(dy, gamma, cache):
    xhat, rstd = cache
    red = (0, 1, 2)
    m = dy.shape[0] * dy.shape[1] * dy.shape[2]
    dgamma = (dy * xhat).sum(axis=red)
    dbeta = dy.sum(axis=red)
    dx = gamma * rstd * (dy - dbeta / m - xhat * dgamma / m)
    return dx, dgamma, dbeta


def batchnorm_infer_fwd(x, gamma, beta, moving_mean, moving_var, eps: float = BN_EPS):
    return (x - moving_mean) / np.sqrt(moving_var + eps) * gamma + beta


def maxpool2_fwd(x):
    n, h, w, c = x.shape
    xr = x[:, : h // 2 * 2, : w // 2 * 2, :].reshape(n, h // 2, 2, w // 2, 2, c)
    return xr.max(axis=(2, 4))


def maxpool2_bwd(dy, x):
    """Gradient goes to the FIRST maximal element of each 2x2 window (TF MaxPoolGrad)."""
    n, h, w, c = x.shape
    oh, ow = h // 2, w // 2
    xr = x[:, : oh * 2, : ow * 2, :].reshape(n, oh, 2, ow, 2, c).transpose(0, 1, 3, 5, 2, 4).reshape(n, oh, ow, c, 4)
    idx = xr.argmax(axis=-1)
    g = np.zeros_like(xr)
    np.put_along_axis(g, idx[..., None], dy[..., None], axis=-1)
    dx = np.zeros_like(x)
    dx[:, : oh * 2, : ow * 2, :] = g.reshape(n, oh, ow, c, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(n, oh * 2, ow * 2, c)
    return dx


def upsample2_bilinear_fwd(x):
    """UpSampling2D(2, interpolation='bilinear') = half-pixel bilinear x2 (no antialias needed)."""
    n, h, w, c = x.shape
    return resize_aa_fwd(x, 2 * h, 2 * w)


def conv_transpose2x2s2_fwd(x, w, b):
    """Conv2DTranspose(nf, 2, strides=2); Keras kernel layout [kh, kw, Cout, Cin]."""
    n, h, wd, cin = x.shape
    kh, kw, cout, cin2 = w.shape
    assert (kh, kw) == (2, 2) and cin2 == cin
    y = np.zeros((n, 2 * h, 2 * wd, cout), dtype=x.dtype)
    for a in range(2):
        for bb in range(2):
            y[:, a::2, bb::2, :] = x @ w[a, bb].T
    return y + b


def conv_transpose2x2s2_bwd(x, w, dy):
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    for a in range(2):
        for bb in range(2):
            g = dy[:, a::2, bb::2, :]
            dx += g @ w[a, bb]
            dw[a, bb] = g.reshape(-1, g.shape[-1]).T @ x.reshape(-1, x.shape[-1])
    db = dy.reshape(-1, dy.shape[-1]).sum(axis=0)
    return dx, dw, db


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def bce_from_probs(y_true, p, eps: float = 1e-7):
    """keras.losses.binary_crossentropy on probabilities (clipped to [eps, 1-eps]), mean over all."""
    p = np.clip(p, eps, 1.0 - eps)
    return (-(y_true * np.log(p) + (1.0 - y_true) * np.log(1.0 - p))).mean()


def dice_coefficient(y_true, y_pred, smooth: float = 1e-6):
    """Segmenation/code/train_adaptive_unet.py:258-269 -- per-sample dice, batch mean."""
    yp = np.clip(y_pred, 1e-7, 1.0 - 1e-7)
    ax = (1, 2, 3)
    inter = (y_true * yp).sum(axis=ax)
    denom = y_true.sum(axis=ax) + yp.sum(axis=ax)
    return ((2.0 * inter + smooth) / (denom + smooth)).mean()


def iou_score(y_true, y_pred, smooth: float = 1e-6):
    """Segmenation/code/train_adaptive_unet.py:272-281 (soft IoU on clipped probabilities)."""
    yp = np.clip(y_pred, 1e-7, 1.0 - 1e-7)
    ax = (1, 2, 3)
    inter = (y_true * yp).sum(axis=ax)
    union = (y_true + yp).sum(axis=ax) - inter
    return ((inter + smooth) / (union + smooth)).mean()


def seg_loss_fwd_bwd(y_true, p, bce_weight: float, dice_weight: float, smooth: float = 1e-6):
    """loss = bce_weight * BCE + dice_weight * (1 - dice)  (make_hybrid_ce_dice_loss / make_bce_dice_loss,
    Segmenation/code/train_adaptive_unet.py:283-304) and its gradient w.r.t. the probabilities p."""
    eps = 1e-7
    pc = np.clip(p, eps, 1.0 - eps)
    inside = (p >= eps) & (p <= 1.0 - eps)
    n = p.shape[0]
    ax = (1, 2, 3)
    bce = (-(y_true * np.log(pc) + (1.0 - y_true) * np.log(1.0 - pc))).mean()
    inter = (y_true * pc).sum(axis=ax, keepdims=True)
    den = (y_true + pc).sum(axis=ax, keepdims=True) + smooth
    dice = ((2.0 * inter + smooth) / den)
    loss = bce_weight * bce + dice_weight * (1.0 - dice.mean())
    dbce = (-(y_true / pc) + (1.0 - y_true) / (1.0 - pc)) / p.size
    ddice = (2.0 * y_true * den - (2.0 * inter + smooth)) / (den * den)
    dp = (bce_weight * dbce - dice_weight * ddice / n) * inside
    return loss, dp


def glorot_uniform(rng: np.random.Generator, shape, dtype=np.float32) -> np.ndarray:
    """Keras GlorotUniform for a conv kernel [kh, kw, cin, cout]."""
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(dtype)
