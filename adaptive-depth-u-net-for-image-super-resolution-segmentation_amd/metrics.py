"""Evaluation metrics of the reference's eval loops: device kernels (csrc/metrics.hip) for tensors that live in HBM
(`DeviceMetrics`, used by evaluate_model.evaluate) and the host-side definitions the evaluator needs (luma for host arrays,
shave rule, PSNR from a per-image MSE, aggregation).  The NumPy restatement of SSIM / MS-SSIM the kernels are tested against
lives in oracle/metrics.py (test infrastructure).

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


def psnr_from_mse(mse: np.ndarray, max_val: float = 1.0) -> np.ndarray:
    """tf.image.psnr's float32 arithmetic on a per-image MSE: 20 log(max_val)/log(10) - float32(10/ln 10) * ln(mse), `inf` at
    MSE 0.  This form (not -10 * log10) is what reproduces the `psnr_y` column of the reference's committed
    per_image_metrics.csv files from their `mse_y` column (tests/test_reference_metric_reports.py)."""
    mse = np.asarray(mse, np.float32)
    with np.errstate(divide="ignore"):
        head = np.float32(20.0) * np.log(np.float32(max_val)) / np.log(np.float32(10.0))
        return (head - np.float32(10.0 / np.log(10.0)) * np.log(mse)).astype(np.float32)


def aggregate(values) -> tuple[float, float]:
    """evaluate_model.py:141-143 `stats`: float64 mean and population std of the per-patch float32 values (one `inf`
    gives mean inf / std nan, as in the reference's scale-0.20 reports)."""
    arr = np.asarray(values).astype(np.float64)
    with np.errstate(invalid="ignore"):
        return float(np.mean(arr)), float(np.std(arr))


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
