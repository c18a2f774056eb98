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
