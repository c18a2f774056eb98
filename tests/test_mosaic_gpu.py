"""The image mosaic of the wave-specialised conv kernels (csrc/conv.hip, Geo / plan_mosaic): on maps whose extent is not a
multiple of the 16 x 16 tile -- the 56- and 34-wide levels of the scale-0.6 pyramid, odd sizes like the 89 / 63 / 45-wide ones of scale 0.7
(Super_resolution/sbatch_scripts/run_experiment_adaptive_depth.sh:47-55 with train_adaptive_unet.py:245-262) -- the tiles walk
ONE virtual map of all images with a single zero line between neighbours.  Only addresses change, so

* forward / dgrad results must be BITWISE those of the per-image tiling (library option "no_mosaic"),
* weight gradients sum the same products in another tile order: equal to fp32 summation accuracy, bitwise deterministic,
* and both are checked against the float64 oracle like every other conv launch (tolerances: tests/test_ops_gpu.py).

The planner takes the mosaic only where it saves a whole ROUND of the persistent kernels (256 workgroups, each one tile of one
64-channel block at a time): test_the_mosaic_is_taken_where_it_saves_a_round.
"""
import numpy as np
import pytest
import torch

from oracle import ops as ref

pytestmark = pytest.mark.gpu

F32, BF16, F16 = torch.float32, torch.bfloat16, torch.float16
TOL = {BF16: 1.5e-2, F16: 2e-3}


def to_dev(a, dtype, device):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device).to(dtype).contiguous()


def rnd(a, dtype):
    return torch.tensor(a, dtype=torch.float32).to(dtype).to(torch.float64).numpy()


def relerr(got, want):
    got = got.detach().to(torch.float64).cpu().numpy()
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))


class per_image_tiling:
    """`with per_image_tiling():` -- the library's A/B switch, restored on exit."""

    def __enter__(self):
        from adunet_amd import _lib
        self.lib = _lib.load()
        self.lib.ad_set_option(b"no_mosaic", 1)

    def __exit__(self, *exc):
        self.lib.ad_set_option(b"no_mosaic", 0)


def mosaic_row(shape, dtype, wgrad=0):
    from adunet_amd import _lib, ops
    n, h, w, c1, c2, cout = shape
    return _lib.load().ad_conv3x3_mosaic(n, h, w, c1, c2, cout, ops.dt(dtype), wgrad)


# n, h, w, c1, c2, cout, images per mosaic row the planner must pick
CASES = [
    (32, 34, 34, 128, 0, 128, None),     # the deepest level of the scale-0.6 pyramid (Experiment 2): 288 tiles -> 162
    (11, 34, 34, 128, 0, 512, 3),        # 11 images, 3 per row: the last mosaic row holds TWO (99 tiles -> 63)
    (9, 56, 56, 64, 64, 256, None),      # virtual concat of two inputs; 144 tiles -> 121
    (15, 43, 88, 128, 0, 128, 3),        # h != w, 3 x 5 images; 270 tiles -> 238 (3 rounds of 128 tiles -> 2)
    (1000, 17, 17, 128, 0, 64, None),    # many small maps one pixel past a tile: 4 000 tiles -> ~1 300, a 18-pixel pitch
]


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("case", CASES)
def test_forward_on_the_mosaic_is_bitwise_the_per_image_tiling_and_matches_the_oracle(device, dtype, case):
    from adunet_amd import ops
    shape, want_row = case[:6], case[6]
    n, h, w, c1, c2, cout = shape
    row = mosaic_row(shape, dtype)
    assert row > 0, "this shape must take the mosaic"
    if want_row:
        assert row == want_row
    rng = np.random.default_rng(sum(shape))
    x = rnd(rng.standard_normal((n, h, w, c1 + c2)), dtype)
    wk = rnd(rng.standard_normal((3, 3, c1 + c2, cout)) * 0.05, dtype)
    b = rng.standard_normal(cout).astype(np.float32).astype(np.float64)
    x1 = to_dev(x[..., :c1], dtype, device)
    x2 = to_dev(x[..., c1:], dtype, device) if c2 else None
    wf, _ = ops.conv3x3_pack(torch.tensor(wk, dtype=F32, device=device), c1 + c2, dtype, want_dgrad=False)
    bias = torch.tensor(b, dtype=F32, device=device)
    y = ops.conv3x3_fwd(x1, x2, wf, bias, cout, relu=False)
    yr = ops.conv3x3_fwd(x1, x2, wf, bias, cout, relu=True)
    with per_image_tiling():
        assert mosaic_row(shape, dtype) == 0
        y0 = ops.conv3x3_fwd(x1, x2, wf, bias, cout, relu=False)
        yr0 = ops.conv3x3_fwd(x1, x2, wf, bias, cout, relu=True)
    assert torch.equal(y, y0) and torch.equal(yr, yr0)
    want = ref.conv2d_same_fwd(x, wk, b)
    assert relerr(y, want) < TOL[dtype]
    assert relerr(yr, np.maximum(want, 0)) < TOL[dtype]


@pytest.mark.parametrize("dtype", [BF16, F16])
def test_dgrad_with_split_outputs_on_the_mosaic(device, dtype):
    """dgrad of a decoder conv (two inputs of 128 channels): the same kernel on the rotated pack, the result split over two
    tensors on a 64-channel block."""
    from adunet_amd import ops
    n, h, w, cin, cout = 32, 34, 34, 256, 128
    assert mosaic_row((n, h, w, cout, 0, cin), dtype) > 0
    rng = np.random.default_rng(3)
    wk = rnd(rng.standard_normal((3, 3, cin, cout)) * 0.05, dtype)
    dz = rnd(rng.standard_normal((n, h, w, cout)), dtype)
    _, wd = ops.conv3x3_pack(torch.tensor(wk, dtype=F32, device=device), cin, dtype)
    dzd = to_dev(dz, dtype, device)
    d1, d2 = ops.conv3x3_fwd(dzd, None, wd, None, cin, split=128)
    with per_image_tiling():
        e1, e2 = ops.conv3x3_fwd(dzd, None, wd, None, cin, split=128)
    assert torch.equal(d1, e1) and torch.equal(d2, e2)
    want, _, _ = ref.conv2d_same_bwd(np.zeros((n, h, w, cin)), wk, dz)
    assert relerr(d1, want[..., :128]) < TOL[dtype] and relerr(d2, want[..., 128:]) < TOL[dtype]


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("case", CASES)
def test_weight_gradient_on_the_mosaic(device, ws, dtype, case):
    from adunet_amd import ops
    shape = case[:6]
    n, h, w, c1, c2, cout = shape
    cin = c1 + c2
    assert mosaic_row(shape, dtype, wgrad=1) > 0, "this shape must take the mosaic"
    rng = np.random.default_rng(sum(shape) + 1)
    x = rnd(rng.standard_normal((n, h, w, cin)), dtype)
    dz = rnd(rng.standard_normal((n, h, w, cout)), dtype)
    x1 = to_dev(x[..., :c1], dtype, device)
    x2 = to_dev(x[..., c1:], dtype, device) if c2 else None
    dzd = to_dev(dz, dtype, device)
    dw = torch.full((3, 3, cin, cout), float("nan"), dtype=F32, device=device)
    ops.conv3x3_wgrad(x1, x2, dzd, dw, cin, ws)
    dw2 = torch.full((3, 3, cin, cout), float("nan"), dtype=F32, device=device)
    ops.conv3x3_wgrad(x1, x2, dzd, dw2, cin, ws)
    assert torch.equal(dw, dw2), "wgrad must be bitwise deterministic"
    with per_image_tiling():
        assert mosaic_row(shape, dtype, wgrad=1) == 0
        dw0 = torch.full((3, 3, cin, cout), float("nan"), dtype=F32, device=device)
        ops.conv3x3_wgrad(x1, x2, dzd, dw0, cin, ws)
    assert float((dw - dw0).abs().max() / dw0.abs().max()) < 1e-5      # the same products, another fp32 summation order
    _, want, _ = ref.conv2d_same_bwd(x, np.zeros((3, 3, cin, cout)), dz, need_dx=False)
    assert relerr(dw, want) < 1e-3


def test_maps_that_are_whole_tiles_or_single_images_keep_the_per_image_tiling(device):
    for shape in [(64, 64, 64, 128, 0, 128), (32, 128, 128, 128, 0, 128), (1, 154, 154, 128, 0, 128), (64, 256, 256, 64, 0, 64)]:
        assert mosaic_row(shape, BF16) == 0 and mosaic_row(shape, BF16, wgrad=1) == 0, shape
    # float32 runs on the generic kernels
    assert mosaic_row((32, 34, 34, 128, 0, 128), F32) == 0


def test_the_mosaic_is_taken_where_it_saves_a_round(device):
    """Scale 0.7, batch 8 (Experiment 2's deepest run), its 45-wide level: 72 tiles x 32 blocks = 9 rounds image by image, 69 x 32
    = 8.6 -> still 9 on the best mosaic: not taken.  The same level at batch 32: 36 rounds -> 35: taken."""
    assert mosaic_row((8, 45, 45, 2048, 0, 2048), BF16) == 0
    assert mosaic_row((32, 45, 45, 2048, 0, 2048), BF16) > 0
    # scale 0.6, batch 32 (run_experiment_adaptive_depth.sh:47-55): the 56- and 34-wide levels
    assert mosaic_row((32, 56, 56, 512, 0, 512), BF16) > 0 and mosaic_row((32, 34, 34, 1024, 0, 1024), BF16) > 0
    assert mosaic_row((32, 56, 56, 512, 0, 512), BF16, wgrad=1) > 0 and mosaic_row((32, 34, 34, 1024, 0, 1024), BF16, wgrad=1) > 0


def test_batches_of_2gib_and_more_take_the_mosaic_run_by_run(device, ws):
    """Tensors that reach 2 GiB are cut into runs of images inside the library (32-bit buffer offsets); every run plans its own
    mosaic.  1 100 images of 34 x 34 x 1 024 (2.6 GB): forward bitwise the per-image tiling, wgrad equal to summation order."""
    from adunet_amd import ops
    n, h, w, c = 1100, 34, 34, 1024
    g = torch.Generator(device=device).manual_seed(7)
    x = (torch.rand((n, h, w, c), device=device, generator=g) - 0.5).bfloat16()
    wk = (torch.rand((3, 3, c, 128), device=device, generator=g) - 0.5) * 0.05
    wf, _ = ops.conv3x3_pack(wk, c, BF16, want_dgrad=False)
    b = torch.rand(128, device=device, generator=g)
    y = ops.conv3x3_fwd(x, None, wf, b, 128)
    dz = y
    dw = torch.empty((3, 3, c, 128), dtype=F32, device=device)
    ops.conv3x3_wgrad(x, None, dz, dw, c, ws)
    with per_image_tiling():
        y0 = ops.conv3x3_fwd(x, None, wf, b, 128)
        dw0 = torch.empty_like(dw)
        ops.conv3x3_wgrad(x, None, dz, dw0, c, ws)
    assert torch.equal(y, y0)
    assert float((dw - dw0).abs().max() / dw0.abs().max()) < 1e-5
    # the query follows the launch path (image runs planned one by one): it is asked about the whole batch
    assert mosaic_row((n, h, w, c, 0, 128), BF16) > 0 and mosaic_row((n, h, w, c, 0, 128), BF16, wgrad=1) > 0
