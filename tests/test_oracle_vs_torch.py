"""The oracle has no TensorFlow to be pinned against (parity unpinned: see oracle/__init__.py), so every op is
cross-checked here against an independent second implementation, PyTorch-CPU, in float64 -- with the known
TF-vs-torch deltas (LN eps 1e-3, HWIO kernels, Keras Adam epsilon placement) encoded explicitly."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops
from oracle.sr_unet import SRUNetOracle

RNG = np.random.default_rng(0)


def t(a):
    return torch.tensor(a, dtype=torch.float64, requires_grad=True)


def nchw(x):
    return x.permute(0, 3, 1, 2)


def test_conv3x3_fwd_bwd():
    x = RNG.standard_normal((2, 9, 7, 5))
    w = RNG.standard_normal((3, 3, 5, 6))
    b = RNG.standard_normal(6)
    dy = RNG.standard_normal((2, 9, 7, 6))
    xt, wt, bt = t(x), t(w), t(b)
    yt = F.conv2d(nchw(xt), wt.permute(3, 2, 0, 1), bt, padding=1).permute(0, 2, 3, 1)
    yt.backward(torch.tensor(dy))
    assert np.allclose(ops.conv2d_same_fwd(x, w, b), yt.detach().numpy(), atol=1e-12)
    dx, dw, db = ops.conv2d_same_bwd(x, w, dy)
    assert np.allclose(dx, xt.grad.numpy(), atol=1e-12) and np.allclose(dw, wt.grad.numpy(), atol=1e-11)
    assert np.allclose(db, bt.grad.numpy(), atol=1e-12)


def test_layernorm_eps_1e3():
    x = RNG.standard_normal((2, 3, 4, 16)) * 0.01   # small variance makes eps matter
    g = RNG.uniform(0.5, 1.5, 16)
    b = RNG.standard_normal(16)
    dy = RNG.standard_normal(x.shape)
    xt, gt, bt = t(x), t(g), t(b)
    yt = F.layer_norm(xt, (16,), gt, bt, eps=1e-3)
    yt.backward(torch.tensor(dy))
    y, cache = ops.layernorm_fwd(x, g, b)
    dx, dg, db = ops.layernorm_bwd(dy, g, cache)
    assert np.allclose(y, yt.detach().numpy(), atol=1e-12)
    assert np.allclose(dx, xt.grad.numpy(), atol=1e-10) and np.allclose(dg, gt.grad.numpy(), atol=1e-10)
    assert np.allclose(db, bt.grad.numpy(), atol=1e-12)
    assert not np.allclose(y, F.layer_norm(t(x), (16,), t(g), t(b), eps=1e-5).detach().numpy(), atol=1e-3)


@pytest.mark.parametrize("sizes", [(37, 23), (32, 16), (40, 10), (10, 40), (23, 37), (8, 2), (5, 1)])
def test_antialiased_resize_matches_torch(sizes):
    i, o = sizes
    x = RNG.standard_normal((2, i, i, 3))
    dy = RNG.standard_normal((2, o, o, 3))
    xt = t(x)
    yt = F.interpolate(nchw(xt), size=(o, o), mode="bilinear", antialias=True, align_corners=False).permute(0, 2, 3, 1)
    yt.backward(torch.tensor(dy))
    assert np.allclose(ops.resize_aa_fwd(x, o, o), yt.detach().numpy(), atol=5e-6)     # float32 tap weights in TF
    assert np.allclose(ops.resize_aa_bwd(dy, i, i), xt.grad.numpy(), atol=5e-6)


@pytest.mark.parametrize("sizes", [(37, 23), (64, 16), (256, 154), (40, 10), (23, 37), (16, 64)])
def test_antialiased_resize_matches_pillow(sizes):
    """A third implementation of the same triangle-filter resize (Pillow's BILINEAR, which widens the support when
    shrinking, as tf.image.resize(antialias=True) does): float32 inside Pillow, hence the 5e-5 bound."""
    Image = pytest.importorskip("PIL.Image")
    i, o = sizes
    x = RNG.standard_normal((i, i)).astype(np.float32)
    want = np.asarray(Image.fromarray(x, mode="F").resize((o, o), resample=Image.BILINEAR))
    got = ops.resize_aa_fwd(x[None, :, :, None].astype(np.float64), o, o)[0, :, :, 0]
    assert np.abs(got - want).max() < 5e-5


def test_upsample_is_plain_half_pixel_bilinear():
    x = RNG.standard_normal((1, 6, 6, 2))
    want = F.interpolate(nchw(t(x)), scale_factor=2, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    assert np.allclose(ops.upsample2_bilinear_fwd(x), want.detach().numpy(), atol=1e-6)


def test_clip_add_and_losses():
    inp = RNG.uniform(-0.3, 1.3, (2, 4, 4, 3))
    res = RNG.standard_normal(inp.shape) * 0.3
    y = RNG.random(inp.shape)
    it, rt = t(inp), t(res)
    out_t = torch.clamp(it + rt, 0.0, 1.0)
    loss_t = torch.sqrt((torch.tensor(y) - out_t) ** 2 + 1e-6).mean()
    loss_t.backward()
    out, pre = ops.clip_add_fwd(inp, res)
    assert np.allclose(out, out_t.detach().numpy())
    assert np.isclose(ops.charbonnier_fwd(y, out), float(loss_t))
    dr = ops.clip_add_bwd(ops.charbonnier_bwd(y, out), pre)
    assert np.allclose(dr, rt.grad.numpy(), atol=1e-12)
    assert np.isclose(ops.l1_fwd(y, out), float((torch.tensor(y) - out_t).abs().mean()))
    assert np.isinf(ops.psnr_per_image(y, y)).all()                                    # MSE 0 -> inf, as tf.image.psnr
    mse = ((y - out) ** 2).reshape(2, -1).mean(1)
    assert np.allclose(ops.psnr_per_image(y, out), -10 * np.log10(mse))


def test_keras_adam_differs_from_torch_adam_only_in_epsilon_placement():
    p0 = RNG.standard_normal(50)
    g = [RNG.standard_normal(50) * 1e-3 for _ in range(4)]
    p, m, v = p0.copy(), np.zeros(50), np.zeros(50)
    pt = torch.tensor(p0.copy(), requires_grad=True)
    opt = torch.optim.Adam([pt], lr=1e-3, eps=0.0)
    for s, gi in enumerate(g, 1):
        ops.adam_step(p, gi, m, v, s, lr=1e-3, eps=0.0)
        pt.grad = torch.tensor(gi)
        opt.step()
    assert np.allclose(p, pt.detach().numpy(), atol=1e-12)       # identical when eps = 0
    p2, m2, v2 = p0.copy(), np.zeros(50), np.zeros(50)
    ops.adam_step(p2, g[0], m2, v2, 1, lr=1e-3, eps=1e-7)
    keras = p0 - 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9) * (0.1 * g[0]) / (np.sqrt(0.001 * g[0] ** 2) + 1e-7)
    assert np.allclose(p2, keras, atol=1e-15)


def test_tier2_ops_match_torch():
    x = RNG.standard_normal((3, 6, 6, 4))
    g = RNG.uniform(0.5, 1.5, 4)
    b = RNG.standard_normal(4)
    dy = RNG.standard_normal(x.shape)
    xt, gt, bt = t(x), t(g), t(b)
    yt = F.batch_norm(nchw(xt), None, None, gt, bt, training=True, eps=1e-3).permute(0, 2, 3, 1)
    yt.backward(torch.tensor(dy))
    y, cache, mu, var = ops.batchnorm_train_fwd(x, g, b)
    dx, dg, db = ops.batchnorm_train_bwd(dy, g, cache)
    assert np.allclose(y, yt.detach().numpy(), atol=1e-10) and np.allclose(dx, xt.grad.numpy(), atol=1e-10)
    assert np.allclose(dg, gt.grad.numpy(), atol=1e-10) and np.allclose(db, bt.grad.numpy(), atol=1e-10)
    xt = t(x)
    pt = F.max_pool2d(nchw(xt), 2).permute(0, 2, 3, 1)
    dp = RNG.standard_normal(tuple(pt.shape))
    pt.backward(torch.tensor(dp))
    assert np.allclose(ops.maxpool2_fwd(x), pt.detach().numpy()) and np.allclose(ops.maxpool2_bwd(dp, x), xt.grad.numpy())
    w = RNG.standard_normal((2, 2, 5, 4))       # Keras layout [kh, kw, Cout, Cin]
    bb = RNG.standard_normal(5)
    xt, wt = t(x), t(w)
    ct = F.conv_transpose2d(nchw(xt), wt.permute(3, 2, 0, 1), torch.tensor(bb), stride=2).permute(0, 2, 3, 1)
    dc = RNG.standard_normal(tuple(ct.shape))
    ct.backward(torch.tensor(dc))
    assert np.allclose(ops.conv_transpose2x2s2_fwd(x, w, bb), ct.detach().numpy(), atol=1e-12)
    dxc, dwc, dbc = ops.conv_transpose2x2s2_bwd(x, w, dc)
    assert np.allclose(dxc, xt.grad.numpy(), atol=1e-12) and np.allclose(dwc, wt.grad.numpy(), atol=1e-11)
    p = RNG.uniform(0.01, 0.99, (2, 4, 4, 1))
    yb = (RNG.random(p.shape) > 0.5).astype(np.float64)
    assert np.isclose(ops.bce_from_probs(yb, p), float(F.binary_cross_entropy(torch.tensor(p), torch.tensor(yb))))


def test_whole_model_gradients_by_torch_autograd():
    """Independent check of the oracle's hand-written backward: the same network rebuilt from torch ops
    (oracle/torch_standin.py, which is also bench.py's PyTorch-CPU baseline)."""
    from oracle.torch_standin import TorchSRUNet
    net = TorchSRUNet(0.6, 2, 20, base_channels=8, head_channels=8, dtype=torch.float64)
    m = net.oracle
    params = m.init_params(np.random.default_rng(3), head_uniform=0.05)
    hr = RNG.random((2, 20, 20, 3))
    lr = np.clip(hr + 0.05 * RNG.standard_normal(hr.shape), 0, 1)
    want_loss, grads, out, _ = m.loss_and_grads(params, lr, hr)
    net.set_params(params)
    loss, o = net.loss(torch.tensor(lr), torch.tensor(hr))
    loss.backward()
    assert np.isclose(float(loss), want_loss, rtol=1e-5)        # resize tap weights are float32 in the oracle (as in TF)
    assert np.abs(o.detach().numpy() - out).max() < 1e-6
    for k in params:
        g = net.P[k].grad.numpy()
        assert np.abs(g - grads[k]).max() <= 2e-5 * max(np.abs(g).max(), 1e-12) + 1e-12, k
    # ... and its Keras-form Adam step against the oracle's
    state = {}
    m.train_step(params, state, lr, hr, lr=1e-3)
    net.set_params({k: v for k, v in m.init_params(np.random.default_rng(3), head_uniform=0.05).items()})
    net.train_step(torch.tensor(lr), torch.tensor(hr), lr=1e-3)
    for k in params:
        assert np.abs(net.P[k].detach().numpy() - params[k]).max() < 2e-5, k


def test_seg_loss_gradient_by_torch_autograd():
    y = (RNG.random((3, 6, 5, 1)) < 0.4).astype(np.float64)
    p = RNG.uniform(0.02, 0.98, y.shape)
    for wb, wd in ((0.4, 0.6), (0.5, 1.0)):
        pt = t(p)
        pc = torch.clamp(pt, 1e-7, 1 - 1e-7)
        yt = torch.tensor(y)
        bce = F.binary_cross_entropy(pc, yt)
        inter = (yt * pc).sum(dim=(1, 2, 3))
        dice = ((2 * inter + 1e-6) / ((yt + pc).sum(dim=(1, 2, 3)) + 1e-6)).mean()
        loss = wb * bce + wd * (1 - dice)
        loss.backward()
        want_loss, dp = ops.seg_loss_fwd_bwd(y, p, wb, wd)
        assert np.isclose(want_loss, float(loss.detach())) and np.allclose(dp, pt.grad.numpy(), atol=1e-12)
        assert np.isclose(ops.dice_coefficient(y, p), float(dice.detach()))
