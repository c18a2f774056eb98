"""The decoder step `dec_up -> Conv2D(nf, 3, same, relu)` (Super_resolution/code/train_adaptive_unet.py:258-259,
shared/custom_layers.py:121-125) in the factored form the product computes it in -- a bank of nine 1x1 convolutions on
the low-resolution map, then an interpolating / shifting gather -- against the reference graph's form
relu(conv3x3(resize(x))) in float64: forward and every gradient, integer and fractional ratios, 1-pixel sources, the
image border (zero padding of the convolution at the high resolution, edge renormalisation of the resize)."""
import numpy as np
import pytest

from oracle import ops as ref

CASES = [  # n, h, H, cin, cout
    (2, 4, 16, 8, 6),      # x4, the K2' pyramid (256/64/16/4/1)
    (1, 1, 4, 8, 4),       # 1 -> 4: every output pixel reads the one source pixel
    (2, 6, 10, 4, 8),      # 0.6 pyramid step (ceil(10 * 0.6) = 6)
    (1, 9, 15, 4, 4),      # odd / odd
    (1, 7, 7, 4, 4),       # identity resize
    (1, 5, 6, 4, 4),       # ratio barely above one
    (1, 2, 3, 4, 4),
]


@pytest.mark.parametrize("case", CASES)
def test_factored_upconv_equals_conv_of_the_resized_tensor(case):
    n, h, hh, cin, cout = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((n, h, h, cin))
    w = rng.standard_normal((3, 3, cin, cout)) * 0.3
    b = rng.standard_normal(cout)
    up = ref.resize_aa_fwd(x, hh, hh)
    want = ref.conv2d_same_fwd(up, w, b)
    got = ref.upconv_gather_fwd(ref.upconv_bank_fwd(x, w), b, hh, hh)
    assert np.abs(got - want).max() < 1e-12 * max(1.0, np.abs(want).max())
    # gradients of sum(relu(.) * r) for a random r
    r = rng.standard_normal(want.shape)
    g = ref.relu_bwd(r, ref.relu_fwd(want))
    dup, dw_want, db_want = ref.conv2d_same_bwd(up, w, g)
    dx_want = ref.resize_aa_bwd(dup, h, h)
    dyb = ref.upconv_gather_bwd(g, h, h)
    dx, dw = ref.upconv_bank_bwd(x, w, dyb)
    assert np.abs(dx - dx_want).max() < 1e-12 * max(1.0, np.abs(dx_want).max())
    assert np.abs(dw - dw_want).max() < 1e-11 * max(1.0, np.abs(dw_want).max())
    assert np.allclose(g.reshape(-1, cout).sum(0), db_want)


def test_gather_is_the_transpose_of_its_backward():
    rng = np.random.default_rng(3)
    y = rng.standard_normal((1, 5, 5, 9, 3))
    g = rng.standard_normal((1, 8, 8, 3))
    lhs = (ref.upconv_gather_fwd(y, None, 8, 8) * g).sum()
    rhs = (y * ref.upconv_gather_bwd(g, 5, 5)).sum()
    assert abs(lhs - rhs) < 1e-10 * abs(lhs)


def test_whole_model_oracle_is_the_same_function_in_factored_form():
    """SRUNetOracle with every decoder up-conv in the factored association (Storage(factored=...), no rounding) returns
    the loss and every gradient of the reference graph's form: the two differ only in where a reduced-precision run rounds."""
    from oracle.sr_unet import SRUNetOracle, Storage
    rng = np.random.default_rng(0)
    for scale, depth, p in ((0.5, 2, 32), (0.6, 3, 40)):
        o = SRUNetOracle(scale, depth, p)
        params = {k: v.astype(np.float64) for k, v in o.init_params(rng, head_uniform=0.05).items()}
        hr = rng.random((2, p, p, 3))
        lr = np.clip(hr + 0.05 * rng.standard_normal(hr.shape), 0, 1)
        l0, g0, out0, _ = o.loss_and_grads(params, lr, hr)
        l1, g1, out1, _ = o.loss_and_grads(params, lr, hr, storage=Storage(None, None, factored=lambda conv: True))
        assert abs(l0 - l1) < 1e-13 and np.abs(out0 - out1).max() < 1e-12
        assert set(g0) == set(g1)
        for k in g0:
            assert np.abs(g0[k] - g1[k]).max() <= 1e-11 * (np.abs(g0[k]).max() + 1e-30), k
