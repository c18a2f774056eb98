"""CPU restatement of the reference's evaluation metrics.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

What the reference computes (all on the Y channel of the clipped prediction, inside the shaved window):
    Super_resolution/code/train_adaptive_unet.py:144-157   rgb_to_luma_bt601
    Super_resolution/code/train_adaptive_unet.py:686-692   tf.image.psnr / ssim / ssim_multiscale, reduce_mean(square(.))
    Super_resolution/code/evaluate_model.py:49-54          infer_eval_shave
    Super_resolution/code/evaluate_model.py:106-126        the same four per-patch metrics in the offline evaluator
    Super_resolution/code/evaluate_model.py:141-163        aggregation: float64 mean and POPULATION std (np.std, ddof 0)

PINNING STATUS.  These are the only numerics of the reference for which its tree holds outputs
(`Super_resolution/experiments/*/evaluation/*/{per_image_metrics.csv,metrics.json}`, condensed to
`tests/golden/eval_reports.{npz,json}` by `tests/golden/make_metrics_fixture.py`):
  * `psnr_from_mse` is pinned: it reproduces all 15 x 3 598 `psnr_y` cells from the `mse_y` cells to float32 rounding
    (>= 98.5 % of them bit for bit, the rest within 2 units in the last place: TensorFlow takes its own float32 `log`,
    and its PSNR op forms the mean squared error a second time), including `inf` at MSE 0.  TensorFlow's form is
        psnr = 20 log(max_val) / log(10) - float32(10 / ln 10) * ln(mse)          (all float32; tf.image.psnr)
    -- NOT -10 * log10(mse), which differs from the reference's cells in 80 % of the rows (by up to 3 units).
  * `aggregate` is pinned: it reproduces every field of all 15 `metrics.json` files from the CSV columns
    (`psnr_mean = inf`, `psnr_std = nan` for the two runs that contain the all-black patch).
  * the degenerate row `inf, 1.0, 1.0, 0.0` pins all four metrics on identical planes.
  * SSIM / MS-SSIM values on ordinary patches stay unpinned against TensorFlow (no image tensors are shipped): they follow
    tf.image.ssim's published definition (11 x 11 Gaussian, sigma 1.5, VALID filtering, K1 = 0.01, K2 = 0.03; MS-SSIM:
    five scales, 2 x 2 average pooling with symmetric padding of odd extents, power factors below) and are cross-checked
    against a direct 2-D Gaussian-window implementation (tests/test_pipeline_cpu.py).
"""
from __future__ import annotations

import numpy as np

MSSSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)       # tf.image.ssim_multiscale's default power_factors


def rgb_to_luma_bt601(image: np.ndarray) -> np.ndarray:
    """train_adaptive_unet.py:144-157 in float32, as the reference runs it (inputs are cast to float32 first)."""
    image = np.asarray(image, dtype=np.float32)
    coeffs = np.array([65.481, 128.553, 24.966], dtype=np.float32)
    y = (image * coeffs).sum(axis=-1, keepdims=True, dtype=np.float32) + np.float32(16.0)
    return np.clip(y / np.float32(255.0), 0.0, 1.0)


def infer_eval_shave(scale: float, explicit: int | None = None) -> int:
    """evaluate_model.py:49-54."""
    if explicit is not None:
        return max(0, int(explicit))
    inv_scale = 1.0 / scale if scale > 0 else 0.0
    scale_factor = int(round(inv_scale)) if inv_scale > 0 else 0
    return 2 * scale_factor if scale_factor > 0 else 0


def mse_per_image(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """tf.reduce_mean(tf.square(a - b), axis=[1, 2, 3]) in float32 (evaluate_model.py:121)."""
    d = np.asarray(a, np.float32) - np.asarray(b, np.float32)
    return (d * d).reshape(d.shape[0], -1).mean(axis=1)


def psnr_from_mse(mse: np.ndarray, max_val: float = 1.0) -> np.ndarray:
    """tf.image.psnr's arithmetic on a float32 MSE (pinned by the reference's CSVs, see the header)."""
    mse = np.asarray(mse, np.float32)
    with np.errstate(divide="ignore"):
        head = np.float32(20.0) * np.log(np.float32(max_val)) / np.log(np.float32(10.0))
        return (head - np.float32(10.0 / np.log(10.0)) * np.log(mse)).astype(np.float32)


def psnr_per_image(a: np.ndarray, b: np.ndarray, max_val: float = 1.0) -> np.ndarray:
    return psnr_from_mse(mse_per_image(a, b), max_val)


def _gauss_kernel(size: int = 11, sigma: float = 1.5) -> np.ndarray:
    x = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    g = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return g / g.sum()


def _filter_valid(x: np.ndarray, g: np.ndarray) -> np.ndarray:
    """Separable VALID correlation over H and W of [N,H,W,C]."""
    k = g.size
    h = sum(g[i] * x[:, i:x.shape[1] - k + 1 + i] for i in range(k))
    return sum(g[i] * h[:, :, i:h.shape[2] - k + 1 + i] for i in range(k))


def ssim_and_cs(a, b, max_val=1.0, k1=0.01, k2=0.03, size=11, sigma=1.5):
    """(ssim, contrast-structure) per image and channel, float64 (tf.image.ssim's `_ssim_per_channel`)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    g = _gauss_kernel(size, sigma)
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    mu_a, mu_b = _filter_valid(a, g), _filter_valid(b, g)
    aa, bb, ab = _filter_valid(a * a, g), _filter_valid(b * b, g), _filter_valid(a * b, g)
    va, vb, cov = aa - mu_a * mu_a, bb - mu_b * mu_b, ab - mu_a * mu_b
    lum = (2 * mu_a * mu_b + c1) / (mu_a * mu_a + mu_b * mu_b + c1)
    cs = (2 * cov + c2) / (va + vb + c2)
    return (lum * cs).mean(axis=(1, 2)), cs.mean(axis=(1, 2))


def ssim_per_image(a, b, max_val: float = 1.0) -> np.ndarray:
    s, _ = ssim_and_cs(a, b, max_val)
    return s.mean(axis=-1).astype(np.float32)


def msssim_per_image(a, b, max_val: float = 1.0, weights=MSSSIM_WEIGHTS) -> np.ndarray:
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    mcs = []
    for i, _ in enumerate(weights):
        s, cs = ssim_and_cs(a, b, max_val)
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


def aggregate(values) -> tuple[float, float]:
    """evaluate_model.py:141-143 `stats`: float64 mean and population standard deviation of the per-patch float32 values
    (an `inf` among them gives mean inf and std nan, which is what the reference's scale-0.20 reports hold)."""
    arr = np.asarray(values).astype(np.float64)
    with np.errstate(invalid="ignore"):
        return float(np.mean(arr)), float(np.std(arr))
