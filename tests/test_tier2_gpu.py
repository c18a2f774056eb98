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
    gw, gb = torch.empty(ch, dtype=F32, device=device), torch.empty(1, dtype=F32, device=device)
    gx = ops.seg_head_bwd(xd, f(wk.reshape(ch)), f(y), prob, sums, gw, gb, wb, wdice, ws)
    assert relerr(gx, dxh) < TOL[dtype] and relerr(gw, dw.reshape(ch)) < 1e-3 and relerr(gb, db) < 1e-3
