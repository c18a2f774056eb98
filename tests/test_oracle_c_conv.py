"""The oracle's C restatement of the convolution (oracle/csrc/conv_ref.c, used for the large float64 cases of the layer-wise
audits) against the oracle's NumPy definition (oracle/ops.py, itself cross-checked against PyTorch-CPU in
tests/test_oracle_vs_torch.py): forward, dgrad, wgrad, bias gradient; 3x3 and 1x1 kernels, ragged register blocks."""
import numpy as np
import pytest

from oracle import ops as ref


@pytest.mark.parametrize("case", [(2, 40, 56, 64, 64, 3), (1, 33, 47, 35, 21, 3), (3, 64, 64, 128, 64, 3), (2, 96, 96, 64, 40, 1)])
def test_c_convolution_equals_the_numpy_definition(case, monkeypatch):
    n, h, w, cin, cout, k = case
    if not ref._conv_c():
        pytest.skip("oracle/_c/liboracle_conv.so is not built (make -C oracle)")
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((n, h, w, cin))
    wt = rng.standard_normal((k, k, cin, cout))
    b = rng.standard_normal(cout)
    g = rng.standard_normal((n, h, w, cout))
    monkeypatch.setattr(ref, "_use_c", lambda *a: True)                 # (these shapes are below the size threshold)
    y1 = ref.conv2d_same_fwd(x, wt, b)
    dx1, dw1, db1 = ref.conv2d_same_bwd(x, wt, g)
    monkeypatch.setattr(ref, "_use_c", lambda *a: False)
    y0 = ref.conv2d_same_fwd(x, wt, b)
    dx0, dw0, db0 = ref.conv2d_same_bwd(x, wt, g)
    for got, want in ((y1, y0), (dx1, dx0), (dw1, dw0), (db1, db0)):
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


def test_c_layernorm_equals_the_numpy_definition(monkeypatch):
    """oracle_ln_fwd / oracle_ln_bwd (fused per-pixel passes) against the NumPy lines of oracle/ops.py."""
    if not ref._conv_c():
        pytest.skip("oracle/_c/liboracle_conv.so is not built (make -C oracle)")
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 17, 19, 96)) * 2 + 0.7
    gamma, beta = rng.uniform(0.5, 1.5, 96), rng.standard_normal(96)
    dy = rng.standard_normal(x.shape)
    monkeypatch.setattr(ref, "_ln_use_c", lambda *a: True)
    y1, (h1, r1) = ref.layernorm_fwd(x, gamma, beta)
    dx1, dg1, db1 = ref.layernorm_bwd(dy, gamma, (h1, r1))
    monkeypatch.setattr(ref, "_ln_use_c", lambda *a: False)
    y0, (h0, r0) = ref.layernorm_fwd(x, gamma, beta)
    dx0, dg0, db0 = ref.layernorm_bwd(dy, gamma, (h0, r0))
    assert r1.shape == r0.shape
    for got, want in ((y1, y0), (h1, h0), (r1, r0), (dx1, dx0), (dg1, dg0), (db1, db0)):
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


def test_c_bf16_round_is_bitwise_the_numpy_definition(monkeypatch):
    if not ref._conv_c():
        pytest.skip("oracle/_c/liboracle_conv.so is not built (make -C oracle)")
    rng = np.random.default_rng(0)
    x = rng.standard_normal(1 << 20) * np.exp(rng.uniform(-40, 40, 1 << 20))
    x[:9] = [np.inf, -np.inf, np.nan, 0.0, -0.0, 1.00390625, 1.01171875, 3.3895313892515355e38, 1e-45]   # ties, overflow edge
    got = ref.bf16_round(x)
    monkeypatch.setattr(ref, "_conv_c", lambda: False)
    assert np.array_equal(got, ref.bf16_round(x), equal_nan=True)
