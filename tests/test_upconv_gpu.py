"""The decoder step `dec_up -> Conv2D(nf, 3, same, relu)` (Super_resolution/code/train_adaptive_unet.py:258-259,
shared/custom_layers.py:121-125) in its factored form (csrc/upconv.hip), entry point by entry point against the oracle:
the 1x1 bank GEMM, the interpolating gather and its transpose, the bank weight gradient, and the whole step against
relu(conv3x3(resize(x))) -- the reference graph's form -- and its gradients."""
import numpy as np
import pytest
import torch

from oracle import ops as ref

pytestmark = pytest.mark.gpu

F32, BF16, F16 = torch.float32, torch.bfloat16, torch.float16
TOL = {F32: 1e-5, BF16: 1.5e-2, F16: 2e-3}


def rnd(a, dtype):
    return torch.tensor(a, dtype=torch.float32).to(dtype).to(torch.float64).numpy()


def to_dev(a, dtype, device):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device).to(dtype).contiguous()


def relerr(got, want):
    got = got.detach().to(torch.float64).cpu().numpy()
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
# pixels, cin, cout.  The last shape gives a workgroup TWO output blocks (125 pixel tiles x 18 blocks): in fp32 the second
# block's accumulator set-up follows the first block's 16-byte stores directly (the store-data hazard noted in pw_gemm_kernel)
# From 2 048 pixels on the 16-bit types take the LDS-tiled persistent kernel (tile 256 x 192 for the bank's 9 Cout columns,
# 256 x 128 for dx's Cin columns; ragged last pixel tile, several tiles per workgroup, 2 .. 18 k-stages).
# r05: widths that are multiples of 256 with K >= 256 take 256 x 256 tiles on the kernel whose two waves per SIMD run a phase apart
# (pw_gemm_pp_kernel): (130999, 256, 64) gives dx 512 tiles of NINE k-stages on 256 workgroups (two tiles each: the LDS buffer
# parity carries over a tile boundary with an odd stage count; ragged last pixel tile; bf16 only, for the suite's time);
# (20999, 512, 256) gives both products.
@pytest.mark.parametrize("shape", [(1, 64, 64), (300, 128, 64), (513, 192, 128), (2000, 64, 64), (31752, 128, 128), (5000, 256, 64),
                                   (70001, 128, 64), (130999, 256, 64), (20999, 512, 256)])
def test_pointwise_gemm(device, dtype, shape):
    """ad_pw_gemm: ragged pixel counts (not a multiple of the 64-pixel wave tile), several k chunks / output blocks."""
    from adunet_amd import _lib, ops
    m, k, cout = shape
    if m > 100000 and dtype != BF16:
        pytest.skip("the two-tiles-per-workgroup case runs in bf16 only")
    if shape in ((130999, 256, 64), (20999, 512, 256)) and dtype != F32:
        assert _lib.load().ad_pw_gemm_tile_channels(m, 9 * cout, k, ops.dt(dtype)) == 256
        assert shape[2] == 64 or _lib.load().ad_pw_gemm_tile_channels(m, k, 9 * cout, ops.dt(dtype)) == 256
    rng = np.random.default_rng(m + k + cout)
    x = rnd(rng.standard_normal((m, k)), dtype)
    w = rnd(rng.standard_normal((3, 3, k, cout)) * 0.1, dtype)
    bf, bd = ops.pw_bank_pack(torch.tensor(w, dtype=F32, device=device), dtype)
    assert ops.pw_supported(m, k, 9 * cout, dtype)
    y = ops.pw_gemm(to_dev(x, dtype, device).view(m, 1, 1, k), bf, 9 * cout)
    want = ref.upconv_bank_fwd(x.reshape(m, 1, 1, k), w).reshape(m, 9 * cout)
    assert relerr(y.view(m, 9 * cout), want) < TOL[dtype]
    dyb = rnd(rng.standard_normal((m, 9 * cout)), dtype)
    dx = ops.pw_gemm(to_dev(dyb, dtype, device).view(m, 1, 1, 9 * cout), bd, k)
    want_dx, want_dw = ref.upconv_bank_bwd(x.reshape(m, 1, 1, k), w, dyb.reshape(m, 1, 1, 9, cout))
    assert relerr(dx.view(m, k), want_dx.reshape(m, k)) < TOL[dtype]
    ws = ops.Workspace(device)
    dw = torch.full((3, 3, k, cout), float("nan"), dtype=F32, device=device)
    ops.upconv_bank_wgrad(to_dev(x, dtype, device).view(m, 1, 1, k), to_dev(dyb, dtype, device).view(m, 1, 1, 9 * cout), dw, ws)
    assert relerr(dw, want_dw) < 1e-3


GATHER = [  # n, h, H, c
    (2, 4, 16, 64),        # x4 (the K2' pyramid), several strips
    (3, 1, 4, 128),        # one source pixel
    (1, 16, 64, 64),
    (2, 6, 10, 64),        # 0.6 pyramid
    (1, 34, 56, 64),       # an Experiment-2 level (run_experiment_adaptive_depth.sh: scale 0.6)
    (1, 9, 15, 64),
    (1, 7, 7, 64),         # identity resize
    (1, 5, 23, 64),        # ratio 4.6: ten transposed taps
    (2, 34, 56, 512),      # two output columns per workgroup, a 36 KB row piece: the 12-slot staging variant
    (1, 6, 10, 256),
    # 2 x row piece + tables beyond the 64 KB default dynamic-LDS limit on the EIGHT-slot variant (until r03 only the
    # 12-slot instantiations had the limit raised).  Scale 0.8 / depth 5, level 2 of the reference's Experiment 2
    # (run_experiment_adaptive_depth.sh:47-55): 132 -> 164 at 256 channels stages 2 x 32 256 bytes in 16 bits;
    # 208 -> 231 at 128 channels is the fp32 shape that crosses the limit
    (1, 132, 164, 256),
    (1, 208, 231, 128),
]


@pytest.mark.parametrize("dtype", [F32, BF16, F16])
@pytest.mark.parametrize("case", GATHER)
def test_gather_forward_and_transpose(device, dtype, case):
    from adunet_amd import ops
    n, h, hh, c = case
    rng = np.random.default_rng(sum(case))
    tab = ops.UpconvTables(h, h, hh, hh, device)
    assert tab.ok
    if not tab.gather_fwd_ok(c, dtype):          # (fp32 with 512 channels: the row piece exceeds the staging window;
        assert dtype == F32 and c >= 512         #  the model then keeps the resize -> conv pair for that level)
        pytest.skip("row piece larger than the kernel stages")
    yb = rnd(rng.standard_normal((n, h, h, 9, c)), dtype)
    b = rng.standard_normal(c).astype(np.float32).astype(np.float64)
    want = ref.upconv_gather_fwd(yb, b, hh, hh)
    bias = torch.tensor(b, dtype=F32, device=device)
    ybd = to_dev(yb.reshape(n, h, h, 9 * c), dtype, device)
    got = ops.upconv_gather_fwd(ybd, bias, tab, relu=False)
    assert relerr(got, want) < TOL[dtype]
    got = ops.upconv_gather_fwd(ybd, bias, tab, relu=True)
    assert relerr(got, np.maximum(want, 0)) < TOL[dtype]
    g = rnd(rng.standard_normal((n, hh, hh, c)), dtype)
    want_b = ref.upconv_gather_bwd(g, h, h).reshape(n, h, h, 9 * c)
    got_b = ops.upconv_gather_bwd(to_dev(g, dtype, device), tab)
    assert relerr(got_b, want_b) < TOL[dtype]


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("case", [(2, 8, 32, 128, 64), (1, 12, 20, 64, 64), (3, 1, 4, 128, 64)])
def test_factored_step_equals_the_reference_graph(device, dtype, case):
    """bank GEMM -> gather against relu(conv3x3(resize(x))), and the three gradients against that graph's, on the GPU."""
    from adunet_amd import ops
    n, h, hh, cin, cout = case
    rng = np.random.default_rng(sum(case))
    x = rnd(rng.standard_normal((n, h, h, cin)), dtype)
    w = rnd(rng.standard_normal((3, 3, cin, cout)) * 0.05, dtype)
    b = rng.standard_normal(cout).astype(np.float32).astype(np.float64)
    up = ref.resize_aa_fwd(x, hh, hh)
    z = ref.conv2d_same_fwd(up, w, b)
    tab = ops.UpconvTables(h, h, hh, hh, device)
    bf, bd = ops.pw_bank_pack(torch.tensor(w, dtype=F32, device=device), dtype)
    xd = to_dev(x, dtype, device)
    yb = ops.pw_gemm(xd, bf, 9 * cout)
    u = ops.upconv_gather_fwd(yb, torch.tensor(b, dtype=F32, device=device), tab, relu=True)
    tol = 1e-4 if dtype == F32 else 2.5e-2           # (bf16: the bank Y is stored in bf16 before the gather)
    assert relerr(u, np.maximum(z, 0)) < tol
    g = rnd(rng.standard_normal(z.shape) * (z > 0), dtype)
    dup, dw_want, _ = ref.conv2d_same_bwd(up, w, g)
    dx_want = ref.resize_aa_bwd(dup, h, h)
    dyb = ops.upconv_gather_bwd(to_dev(g, dtype, device), tab)
    dx = ops.pw_gemm(dyb, bd, cin)
    assert relerr(dx, dx_want) < tol
    dw = torch.empty((3, 3, cin, cout), dtype=F32, device=device)
    ops.upconv_bank_wgrad(xd, dyb, dw, ops.Workspace(device))
    assert relerr(dw, dw_want) < (1e-4 if dtype == F32 else 1e-2)
