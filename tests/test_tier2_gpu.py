"""GPU parity of the tier-2 (segmentation) ops vs the oracle: BatchNorm(+ReLU) train/infer/backward, MaxPool2,
Conv2DTranspose(2, s=2) as pointwise GEMM + pixel shuffle, sigmoid + BCE/Dice head.  Tolerances as in
test_ops_gpu.py (fp32: 1e-3 relative; bf16: 1.5e-2 of the tensor's max magnitude)."""
import numpy as np
import pytest
import torch

from oracle import ops as ref

pytestmark = pytest.mark.gpu
F32, BF16 = torch.float32, torch.bfloat16
TOL = {F32: 1e-3, BF16: 1.5e-2}


def to_dev(a, dtype, device):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device).to(dtype).contiguous()


def rnd(a, dtype):
    return torch.tensor(a, dtype=torch.float32).to(dtype).to(torch.float64).numpy()


def relerr(got, want):
    got = got.detach().to(torch.float64).cpu().numpy()
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("c", [64, 256, 1024])
def test_batchnorm_relu(device, ws, dtype, c):
    from adunet_amd import ops
    rng = np.random.default_rng(c)
    z = rnd(rng.standard_normal((3, 6, 5, c)) * 1.5 + 0.2, dtype)
    gamma = rng.uniform(0.5, 1.5, c).astype(np.float32).astype(np.float64)
    beta = rng.uniform(-0.5, 0.5, c).astype(np.float32).astype(np.float64)
    y, cache, mu, var = ref.batchnorm_train_fwd(z, gamma, beta)
    a = ref.relu_fwd(y)
    f = lambda v: torch.tensor(v, dtype=F32, device=device)
    mm, mv = f(np.zeros(c)), f(np.ones(c))
    zd = to_dev(z, dtype, device)
    got, mean, rstd = ops.batchnorm_relu_fwd_train(zd, f(gamma), f(beta), mm, mv, ws)
    assert relerr(got, a) < TOL[dtype]
    assert relerr(mean, mu) < 1e-4 and relerr(rstd, cache[1].reshape(-1)) < 1e-4
    assert relerr(mm, 0.01 * mu) < 1e-4 and relerr(mv, 0.99 + 0.01 * var) < 1e-5          # Keras moving averages
    dy = rnd(rng.standard_normal(z.shape), dtype)
    dz, dg, db = ref.batchnorm_train_bwd(ref.relu_bwd(dy, a), gamma, cache)
    dgam, dbet = torch.empty(c, dtype=F32, device=device), torch.empty(c, dtype=F32, device=device)
    gz = ops.batchnorm_relu_bwd(to_dev(dy, dtype, device), zd, mean, rstd, f(gamma), f(beta), dgam, dbet, ws)
    assert relerr(gz, dz) < TOL[dtype] and relerr(dgam, dg) < 1e-3 and relerr(dbet, db) < 1e-3
    mmean, mvar = rng.standard_normal(c), rng.uniform(0.5, 2.0, c)
    want = ref.relu_fwd(ref.batchnorm_infer_fwd(z, gamma, beta, mmean, mvar))
    assert relerr(ops.batchnorm_relu_fwd_infer(zd, f(gamma), f(beta), f(mmean), f(mvar)), want) < TOL[dtype]


@pytest.mark.parametrize("dtype", [F32, BF16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 6, 8, 64), (3, 4, 4, 256), (1, 2, 2, 2048)])
def test_batchnorm_pool_and_dbias_fusions_equal_the_separate_launches(device, ws, dtype, shape):
    """r05: ad_batchnorm_relu_pool_fwd_train = ad_batchnorm_relu_fwd_train + ad_maxpool2_fwd (bitwise: the maximum is taken of the
    stored activations), ad_batchnorm_relu_bwd_dbias = ad_batchnorm_relu_bwd + ad_colsum(dz) (dz bitwise, the sums to fp32
    summation order), and both against the oracle."""
    from adunet_amd import ops
    n, h, w, c = shape
    if not ops.batchnorm_pool_supported(torch.empty(shape, dtype=dtype)):
        pytest.skip("wider than 256 channel vectors: the model falls back to the separate pooling launch")
    rng = np.random.default_rng(sum(shape))
    z = rnd(rng.standard_normal(shape) * 1.3 - 0.4, dtype)
    gamma = rng.uniform(0.5, 1.5, c).astype(np.float32).astype(np.float64)
    beta = rng.uniform(-0.5, 0.5, c).astype(np.float32).astype(np.float64)
    f = lambda v: torch.tensor(v, dtype=F32, device=device)
    zd = to_dev(z, dtype, device)
    y0, mean0, rstd0 = ops.batchnorm_relu_fwd_train(zd, f(gamma), f(beta), f(np.zeros(c)), f(np.ones(c)), ws)
    p0 = ops.maxpool2_fwd(y0)
    mm, mv = f(np.zeros(c)), f(np.ones(c))
    y1, p1, mean1, rstd1 = ops.batchnorm_relu_pool_fwd_train(zd, f(gamma), f(beta), mm, mv, ws)
    assert torch.equal(y0, y1) and torch.equal(p0, p1) and torch.equal(mean0, mean1) and torch.equal(rstd0, rstd1)
    y, cache, mu, var = ref.batchnorm_train_fwd(z, gamma, beta)
    tol = {F32: 1e-3, BF16: 1.5e-2, torch.float16: 2e-3}[dtype]
    assert relerr(p1, ref.maxpool2_fwd(ref.relu_fwd(y))) < tol
    assert relerr(mm, 0.01 * mu) < 1e-4 and relerr(mv, 0.99 + 0.01 * var) < 1e-5
    dy = to_dev(rnd(rng.standard_normal(shape), dtype), dtype, device)
    g0, b0, g1, b1, db1 = (torch.empty(c, dtype=F32, device=device) for _ in range(5))
    dz0 = ops.batchnorm_relu_bwd(dy, zd, mean0, rstd0, f(gamma), f(beta), g0, b0, ws)
    db0 = ops.colsum(dz0.view(-1, c), torch.empty(c, dtype=F32, device=device), ws)
    dz1 = ops.batchnorm_relu_bwd(dy, zd, mean0, rstd0, f(gamma), f(beta), g1, b1, ws, dbias=db1)
    assert torch.equal(dz0, dz1) and torch.equal(g0, g1) and torch.equal(b0, b1)
    want_db = dz1.to(torch.float64).sum(dim=(0, 1, 2)).cpu().numpy()
    scale = float(dz1.to(torch.float64).abs().sum(dim=(0, 1, 2)).max()) + 1e-30
    assert np.abs(db1.cpu().numpy() - want_db).max() / scale < 1e-5 and np.abs(db0.cpu().numpy() - want_db).max() / scale < 1e-5


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_batchnorm_statistics_in_one_pass_with_a_large_mean_offset(device, ws, dtype):
    """The batch moments come out of ONE pass over z (r05) as sums of deviations from the first pixel's row.  A channel whose
    mean is 1 000 standard deviations away from zero (where the unshifted E[x^2] - mean^2 loses 10 % of the variance in fp32:
    the r04 defect of the fused LayerNorm epilogue) and one whose FIRST pixel is an outlier 30 deviations from the mean."""
    from adunet_amd import ops
    rng = np.random.default_rng(5)
    c = 64
    z = rng.standard_normal((4, 16, 16, c))
    z[..., 0] = z[..., 0] * 0.01 + 10.0              # mean / std = 1 000
    z[..., 1] = z[..., 1] * 0.5 - 3.0
    z[0, 0, 0, 1] = 12.0                             # an outlier as the shift
    z = rnd(z, dtype)
    gamma, beta = np.ones(c), np.zeros(c)
    f = lambda v: torch.tensor(v, dtype=F32, device=device)
    _, cache, mu, var = ref.batchnorm_train_fwd(z, gamma, beta)
    _, mean, rstd = ops.batchnorm_relu_fwd_train(to_dev(z, dtype, device), f(gamma), f(beta), None, None, ws)
    got_mu, got_rstd = mean.cpu().numpy().astype(np.float64), rstd.cpu().numpy().astype(np.float64)
    assert np.abs(got_mu - mu).max() < 1e-5 * np.abs(mu).max()
    want_rstd = 1.0 / np.sqrt(var + 1e-3)
    assert np.abs(got_rstd / want_rstd - 1.0).max() < 1e-4, np.abs(got_rstd / want_rstd - 1.0).max()


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("hw", [(8, 8), (7, 10)])
def test_maxpool2(device, dtype, hw):
    from adunet_amd import ops
    rng = np.random.default_rng(1)
    x = rnd(rng.standard_normal((2, hw[0], hw[1], 64)), dtype)
    x[0, 0, 0, :8] = x[0, 0, 1, :8] = 5.0                                               # ties: first element wins
    xd = to_dev(x, dtype, device)
    y = ops.maxpool2_fwd(xd)
    assert relerr(y, ref.maxpool2_fwd(x)) == 0.0
    dy = rnd(rng.standard_normal(tuple(y.shape)), dtype)
    assert relerr(ops.maxpool2_bwd(to_dev(dy, dtype, device), xd), ref.maxpool2_bwd(dy, x)) == 0.0
    # the encoder junction: pooling gradient + skip gradient in one pass, summed in fp32 and rounded once
    skip = rnd(rng.standard_normal(x.shape), dtype)
    got = ops.maxpool2_bwd(to_dev(dy, dtype, device), xd, add=to_dev(skip, dtype, device))
    want = rnd(ref.maxpool2_bwd(dy, x) + skip, dtype)
    assert relerr(got, want) == 0.0


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_conv_transpose2x2s2(device, ws, dtype):
    from adunet_amd import ops
    rng = np.random.default_rng(2)
    n, h, w, cin, cout = 2, 5, 6, 128, 64
    x = rnd(rng.standard_normal((n, h, w, cin)), dtype)
    wt = rnd(rng.standard_normal((2, 2, cout, cin)) * 0.1, dtype)
    b = rng.standard_normal(cout).astype(np.float32).astype(np.float64)
    want = ref.conv_transpose2x2s2_fwd(x, wt, b)
    wf, wd, _ = ops.conv_transpose2x2s2_pack(torch.tensor(wt, dtype=F32, device=device), dtype)
    xd = to_dev(x, dtype, device)
    y = ops.conv_transpose2x2s2_fwd(xd, wf, torch.tensor(b, dtype=F32, device=device), cout)
    assert tuple(y.shape) == (n, 2 * h, 2 * w, cout) and relerr(y, want) < TOL[dtype]
    dy = rnd(rng.standard_normal(want.shape), dtype)
    dx, dw, db = ref.conv_transpose2x2s2_bwd(x, wt, dy)
    gw = torch.empty((2, 2, cout, cin), dtype=F32, device=device)
    gb = torch.empty(cout, dtype=F32, device=device)
    gx = ops.conv_transpose2x2s2_bwd(xd, to_dev(dy, dtype, device), wd, gw, gb, ws)
    assert relerr(gx, dx) < TOL[dtype] and relerr(gw, dw) < 1e-3 and relerr(gb, db) < 1e-3


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("weights", [(0.4, 0.6), (0.5, 1.0)])      # protocols A and B (:382-403)
def test_seg_head(device, ws, dtype, weights):
    from adunet_amd import ops
    wb, wdice = weights
    rng = np.random.default_rng(3)
    n, h, w, ch = 3, 12, 9, 64
    xh = rnd(np.maximum(rng.standard_normal((n, h, w, ch)), 0), dtype)
    wk = rng.uniform(-0.3, 0.3, (1, 1, ch, 1)).astype(np.float32).astype(np.float64)
    b = np.array([0.1])
    y = (rng.random((n, h, w, 1)) < 0.3).astype(np.float64)
    logit = ref.conv2d_same_fwd(xh, wk, b)
    p = ref.sigmoid(logit)
    loss, dp = ref.seg_loss_fwd_bwd(y, p, wb, wdice)
    dlogit = dp * p * (1 - p)
    dxh, dw, db = ref.conv2d_same_bwd(xh, wk, dlogit)
    f = lambda v: torch.tensor(v, dtype=F32, device=device)
    xd = to_dev(xh, dtype, device)
    prob, sums = ops.seg_head_fwd(xd, f(wk.reshape(ch)), f(b), f(y), ws)
    assert relerr(prob, p) < 1e-5
    s = sums.double().cpu().numpy()
    dice = ((2 * s[:, 1] + 1e-6) / (s[:, 2] + 1e-6)).mean()
    got_loss = wb * s[:, 0].sum() / y.size + wdice * (1 - dice)
    assert abs(got_loss - loss) < 1e-5 and abs(dice - ref.dice_coefficient(y, p)) < 1e-6
    m = ops.seg_metrics(sums, float(y.size), wb, wdice).double().cpu().numpy()      # the batch values a train step logs, one launch
    assert abs(m[0] - loss) < 1e-5 and abs(m[1] - ref.dice_coefficient(y, p)) < 1e-6 and abs(m[2] - ref.iou_score(y, p)) < 1e-6
    gw, gb = torch.empty(ch, dtype=F32, device=device), torch.empty(1, dtype=F32, device=device)
    gx = ops.seg_head_bwd(xd, f(wk.reshape(ch)), f(y), prob, sums, gw, gb, wb, wdice, ws)
    assert relerr(gx, dxh) < TOL[dtype] and relerr(gw, dw.reshape(ch)) < 1e-3 and relerr(gb, db) < 1e-3
