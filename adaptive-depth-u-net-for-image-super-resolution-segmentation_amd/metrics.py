"""Evaluation metrics of the reference's eval loops: device kernels (csrc/metrics.hip) for tensors that live in HBM
(`*_device`, used by evaluate_model.evaluate) and a NumPy restatement with the same definitions for host arrays.

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


# ----------------------------------------------------------------------------- device path (csrc/metrics.hip)
class DeviceMetrics:
    """Y-channel PSNR / MSE / SSIM / MS-SSIM of a batch that stays in HBM: luma of the clipped prediction and of the
    target, shave as a strided window (no copy), one launch per metric and scale; returns per-image host arrays."""

    def __init__(self, device):
        import torch
        from . import _lib, ops
        self.torch, self.lib, self.check, self.ops = torch, _lib.load(), _lib.check, ops
        self.device = device
        self.ws = ops.Workspace(device, 1 << 20)

    def luma(self, rgb):
        t = self.torch
        assert rgb.dtype == t.float32 and rgb.is_contiguous() and rgb.shape[-1] == 3
        y = t.empty(rgb.shape[:-1], dtype=t.float32, device=rgb.device)
        self.check(self.lib.ad_luma_bt601(rgb.data_ptr(), y.data_ptr(), y.numel(), t.cuda.current_stream().cuda_stream), "ad_luma_bt601")
        return y

    def _window(self, plane, shave):
        n, h, w = plane.shape
        off = (shave * w + shave) * 4
        return plane.data_ptr() + off, h - 2 * shave, w - 2 * shave, h * w, w

    def mse_ssim(self, ya, yb, shave: int = 0, with_ssim: bool = True):
        """(mse[n], ssim[n] or None, cs[n] or None) of two [n, h, w] fp32 planes inside the shaved window."""
        t = self.torch
        n = ya.shape[0]
        pa, h, w, istr, ld = self._window(ya, shave)
        pb = self._window(yb, shave)[0]
        self.ws.ensure(self.lib.ad_metrics_ws_bytes(n, h, w))
        st = t.cuda.current_stream().cuda_stream
        mse = t.empty(n, dtype=t.float32, device=ya.device)
        self.check(self.lib.ad_mse_per_image(pa, pb, n, h, w, istr, ld, mse.data_ptr(), self.ws.ptr, self.ws.nbytes, st), "ad_mse_per_image")
        if not with_ssim or min(h, w) < 11:
            return mse, None, None
        sc = t.empty((n, 2), dtype=t.float32, device=ya.device)
        self.check(self.lib.ad_ssim_per_image(pa, pb, n, h, w, istr, ld, 1.0, sc.data_ptr(), self.ws.ptr, self.ws.nbytes, st), "ad_ssim_per_image")
        return mse, sc[:, 0], sc[:, 1]

    def msssim(self, ya, yb, shave: int = 0, weights=MSSSIM_WEIGHTS):
        """tf.image.ssim_multiscale on the shaved window: per-image float32 host array."""
        t = self.torch
        n = ya.shape[0]
        pa, h, w, istr, ld = self._window(ya, shave)
        pb = self._window(yb, shave)[0]
        st = t.cuda.current_stream().cuda_stream
        keep = [ya, yb]
        mcs = []
        for i in range(len(weights)):
            self.ws.ensure(self.lib.ad_metrics_ws_bytes(n, h, w))
            sc = t.empty((n, 2), dtype=t.float32, device=ya.device)
            self.check(self.lib.ad_ssim_per_image(pa, pb, n, h, w, istr, ld, 1.0, sc.data_ptr(), self.ws.ptr, self.ws.nbytes, st), "ad_ssim_per_image")
            mcs.append(sc[:, 0] if i == len(weights) - 1 else sc[:, 1])
            if i < len(weights) - 1:
                oh, ow = (h + 1) // 2, (w + 1) // 2
                na = t.empty((n, oh, ow), dtype=t.float32, device=ya.device)
                nb = t.empty_like(na)
                self.check(self.lib.ad_avgpool2_plane(pa, n, h, w, istr, ld, na.data_ptr(), st), "ad_avgpool2_plane")
                self.check(self.lib.ad_avgpool2_plane(pb, n, h, w, istr, ld, nb.data_ptr(), st), "ad_avgpool2_plane")
                keep += [na, nb]
                pa, pb, h, w, istr, ld = na.data_ptr(), nb.data_ptr(), oh, ow, oh * ow, ow
        host = np.maximum(np.stack([m.cpu().numpy().astype(np.float64) for m in mcs], axis=-1), 0.0)     # [n, scales]
        return np.prod(host ** np.asarray(weights), axis=-1).astype(np.float32)
