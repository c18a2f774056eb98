"""Evaluation metrics of the reference's eval loops (host side, NumPy on the model's fp32 outputs).

rgb_to_luma_bt601: Super_resolution/code/train_adaptive_unet.py:144-157; shave: evaluate_model.py:49-54;
PSNR / MSE / SSIM / MS-SSIM on Y: evaluate_model.py:106-126 (tf.image.psnr / ssim / ssim_multiscale).
SSIM follows TensorFlow's definition (11x11 Gaussian, sigma 1.5, VALID filtering, K1=0.01, K2=0.03; MS-SSIM: 5
scales, 2x2 average pooling, the standard power factors); without TensorFlow here it is parity unpinned.
"""
from __future__ import annotations

import numpy as np

MSSSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def rgb_to_luma_bt601(image: np.ndarray) -> np.ndarray:
    image = np.asarray(image, dtype=np.float32)
    coeffs = np.array([65.481, 128.553, 24.966], dtype=np.float32)
    y = (image * coeffs).sum(axis=-1, keepdims=True, dtype=np.float32) + np.float32(16.0)
    return np.clip(y / np.float32(255.0), 0.0, 1.0)


def infer_eval_shave(scale: float, explicit: int | None = None) -> int:
    if explicit is not None:
        return max(0, int(explicit))
    inv_scale = 1.0 / scale if scale > 0 else 0.0
    scale_factor = int(round(inv_scale)) if inv_scale > 0 else 0
    return 2 * scale_factor if scale_factor > 0 else 0


def mse_per_image(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    d = np.asarray(a, np.float32) - np.asarray(b, np.float32)
    return (d * d).reshape(d.shape[0], -1).mean(axis=1)


def psnr_per_image(a: np.ndarray, b: np.ndarray, max_val: float = 1.0) -> np.ndarray:
    with np.errstate(divide="ignore"):
        return (20.0 * np.log10(max_val) - 10.0 * np.log10(mse_per_image(a, b))).astype(np.float32)


def _gauss_kernel(size: int = 11, sigma: float = 1.5) -> np.ndarray:
    x = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    g = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return g / g.sum()


def _filter_valid(x: np.ndarray, g: np.ndarray) -> np.ndarray:
    """Separable VALID correlation over H and W of [N,H,W,C]."""
    k = g.size
    h = sum(g[i] * x[:, i:x.shape[1] - k + 1 + i] for i in range(k))
    return sum(g[i] * h[:, :, i:h.shape[2] - k + 1 + i] for i in range(k))


def _ssim_cs(a, b, max_val=1.0, k1=0.01, k2=0.03, size=11, sigma=1.5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    g = _gauss_kernel(size, sigma)
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    mu_a, mu_b = _filter_valid(a, g), _filter_valid(b, g)
    aa, bb, ab = _filter_valid(a * a, g), _filter_valid(b * b, g), _filter_valid(a * b, g)
    va, vb, cov = aa - mu_a * mu_a, bb - mu_b * mu_b, ab - mu_a * mu_b
    lum = (2 * mu_a * mu_b + c1) / (mu_a * mu_a + mu_b * mu_b + c1)
    cs = (2 * cov + c2) / (va + vb + c2)
    return (lum * cs).mean(axis=(1, 2)), cs.mean(axis=(1, 2))      # per image, per channel


def ssim_per_image(a, b, max_val: float = 1.0) -> np.ndarray:
    s, _ = _ssim_cs(a, b, max_val)
    return s.mean(axis=-1).astype(np.float32)


def msssim_per_image(a, b, max_val: float = 1.0, weights=MSSSIM_WEIGHTS) -> np.ndarray:
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    mcs = []
    for i, _ in enumerate(weights):
        s, cs = _ssim_cs(a, b, max_val)
        mcs.append(np.maximum(s if i == len(weights) - 1 else cs, 0.0))
        if i < len(weights) - 1:
            n, h, w, c = a.shape
            pad_h, pad_w = h % 2, w % 2
            if pad_h or pad_w:                                  # tf pads by symmetric replication before pooling
                a = np.pad(a, ((0, 0), (0, pad_h), (0, pad_w), (0, 0)), mode="symmetric")
                b = np.pad(b, ((0, 0), (0, pad_h), (0, pad_w), (0, 0)), mode="symmetric")
                h, w = h + pad_h, w + pad_w
            a = a.reshape(n, h // 2, 2, w // 2, 2, c).mean(axis=(2, 4))
            b = b.reshape(n, h // 2, 2, w // 2, 2, c).mean(axis=(2, 4))
    mcs = np.stack(mcs, axis=-1)                                 # [N, C, scales]
    return np.prod(mcs ** np.asarray(weights), axis=-1).mean(axis=-1).astype(np.float32)
