"""GPU parity of the segmentation model family (tier 2) vs the NumPy oracle: forward probabilities, loss, dice / IoU,
every gradient tensor, BatchNorm moving statistics and inference mode, Adam steps with the protocol-A cosine schedule."""
import numpy as np
import pytest
import torch

from oracle import ops as ref
from oracle.seg_unet import SegUNetOracle

pytestmark = pytest.mark.gpu


def rel(got, want):
    return float(np.abs(np.asarray(got, np.float64) - want).max() / (np.abs(want).max() + 1e-30))


def storage_of(model, n):
    """Oracle-side description of where `model` rounds (see tests/test_model_gpu.py::storage_of)."""
    from adunet_amd import _lib, ops
    from oracle.sr_unet import Storage
    if model.dtype == torch.float32:
        return None
    lib = _lib.load()
    fused = {}
    for bi, blk in enumerate(model.blocks):
        for i, (cs, nn) in enumerate(blk):
            if model.norm == "bn":
                fused[cs.name] = False
                continue
            decoder_first = bi > model.depth and i == 0          # virtual concat [up | skip]
            c1, c2 = (cs.cin // 2, cs.cin // 2) if decoder_first else (model._cin_pad(cs), 0)
            fused[cs.name] = bool(lib.ad_conv3x3_ln_relu_is_fused(n, cs.hw, cs.hw, c1, c2, cs.cout, ops.dt(model.dtype)))
            if bi == 0 and i == 0 and lib.ad_conv3x3_c3_supported(n, cs.hw, cs.hw, cs.cout, ops.dt(model.dtype)):
                fused[cs.name] = True        # the raw image goes through the 3-channel kernel: conv + LayerNorm in one launch
    return Storage(ref.bf16_round, lambda conv, *shape: fused[conv])


def build(kind, dtype, device, p=32, depth=2, batch=3):
    from adunet_amd import seg_model as S
    if kind == "bn":
        model = S.build_adaptive_depth_unet(p, 64, depth, dtype=dtype, device=device)
        oracle = SegUNetOracle(p, 64, depth, "bn", "bilinear")
    else:
        model = S.build_unet(p, 1, 64, depth, dtype=dtype, device=device)
        oracle = SegUNetOracle(p, 64, depth, "ln", "convT")
    rng = np.random.default_rng(11)
    params, state = oracle.init_params(rng)
    params = {k: v.astype(np.float32).astype(np.float64) for k, v in params.items()}
    assert list(model.index) == list(oracle.param_shapes) and list(model.state_index) == list(oracle.state_shapes)
    model.set_weights({**{k: v.astype(np.float32) for k, v in params.items()}, **{k: v.astype(np.float32) for k, v in state.items()}})
    img = rng.random((batch, p, p, 3), dtype=np.float32)
    mask = (rng.random((batch, p, p, 1)) < 0.35).astype(np.float32)
    return S, model, oracle, params, state, img, mask


def check_seg_step(S, model, oracle, params, state, img, mask, *, f32, kind, deep=False):
    """End-to-end forward / loss / metrics / gradients against the oracle (bf16: its storage mode).

    fp32: 1e-3 on outputs, 2e-3 per gradient tensor; a ReLU pre-activation within float32 rounding of zero may land
    on either side (K3 at batch 2 has ~60 of them among 1.2e8) and BatchNorm spreads one flipped sample over its whole
    channel -- the oracle brackets that (`kink_slack`) and the bracket is added to the bound of the tensors it reaches.
    Conv biases in front of BatchNorm have a mathematically zero gradient and are left out.
    bf16: the rounding-flip cascade described in tests/test_layerwise_gpu.py (where every kernel of these models is
    checked step by step at single-kernel tolerance) makes two bf16 evaluations of a deep BatchNorm net differ like two
    noise draws: the end-to-end check bounds that noise statistically (mean probability error, loss, dice / IoU, flat
    gradient cosine); per-tensor gradient bounds are kept for the LayerNorm models, whose noise is an order smaller."""
    proto = S.PROTOCOLS["A"]
    loss_obj = proto.loss_builder()
    model.compile(optimizer=S.build_optimizer(proto, steps_per_epoch=10, epochs=2), loss=loss_obj)
    st = dict(state)
    storage = storage_of(model, img.shape[0])
    want_loss, grads, p, dice, iou = oracle.loss_and_grads(params, st, img.astype(np.float64), mask.astype(np.float64), 0.4, 0.6,
                                                           storage=storage)
    x, m = model._to_dev(img), model._to_dev_mask(mask)
    prob, sums, tape = model._forward_seg(x, m, training=True, keep=True)
    model._backward_seg(tape, m)
    perr = np.abs(prob.cpu().numpy() - p)
    assert perr.max() < (1e-3 if f32 else 0.15 if deep else 2e-2) and perr.mean() < (1e-5 if f32 else 2e-2 if deep else 3e-3)
    loss, d, i = model._metrics_from(sums, float(m.numel()))
    assert abs(float(loss) - want_loss) < (1e-3 if f32 else 5e-3) * want_loss
    assert abs(float(d) - dice) < (1e-4 if f32 else 5e-3) and abs(float(i) - iou) < (1e-4 if f32 else 5e-3)
    got = model.get_grads()
    live = [k for k in grads if np.abs(grads[k]).max() > 1e-9 and not (kind == "bn" and k.endswith("/bias") and k.startswith("conv2d"))]
    errs = {k: rel(got[k], grads[k]) for k in live}
    if f32:
        bound = {k: 2e-3 for k in live}
        if any(errs[k] >= bound[k] for k in live):
            slack = oracle.kink_slack(params)
            bound = {k: bound[k] + 4.0 * slack[k] / (np.abs(grads[k]).max() + 1e-30) for k in live}
        worst = max((errs[k] / bound[k], k, errs[k], bound[k]) for k in live)
        assert worst[0] < 1.0, worst
    elif kind != "bn":
        worst = max((errs[k], k) for k in live)
        assert worst[0] < 5e-2, worst
    ga = np.concatenate([got[k].reshape(-1) for k in live]).astype(np.float64)
    gb = np.concatenate([grads[k].reshape(-1) for k in live])
    cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
    assert cos > (0.9999 if f32 else 0.8 if deep else 0.95 if kind == "bn" else 0.999), cos
    if kind == "bn":
        w = model.get_weights()
        for k in state:                                  # Keras moving averages after one training batch
            assert rel(w[k], st[k]) < (1e-4 if f32 else 2e-2), k
        want_inf = oracle.forward(params, st, img.astype(np.float64), training=False, storage=storage)
        inf_err = np.abs(model(img, training=False) - want_inf)
        assert inf_err.max() < (1e-3 if f32 else 0.15 if deep else 3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["bn", "ln"])
def test_seg_forward_loss_gradients(device, dtype, kind):
    S, model, oracle, params, state, img, mask = build(kind, dtype, device)
    check_seg_step(S, model, oracle, params, state, img, mask, f32=dtype == torch.float32, kind=kind)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_k3_config_against_oracle(device, dtype):
    """BASELINE config 3 as the reference can express it: build_adaptive_depth_unet(256, 64, 5), batch 2."""
    S, model, oracle, params, state, img, mask = build("bn", dtype, device, p=256, depth=5, batch=2)
    assert model.name == "adaptive_unet_depth5_c64"
    check_seg_step(S, model, oracle, params, state, img, mask, f32=dtype == torch.float32, kind="bn", deep=True)


def test_k3_full_batch_properties(device):
    """K3 at its protocol-A batch of 8: deterministic, hipGraph replay == eager bit for bit, loss finite and falling."""
    from adunet_amd import seg_model as S
    rng = np.random.default_rng(5)
    img = rng.random((8, 256, 256, 3), dtype=np.float32)
    mask = (rng.random((8, 256, 256, 1)) < 0.3).astype(np.float32)
    proto = S.PROTOCOLS["A"]
    finals = []
    for mode in ("eager", "eager", "graph"):
        model = S.build_adaptive_depth_unet(256, 64, 5, dtype=torch.bfloat16, device=device, seed=3)
        model.compile(optimizer=S.build_optimizer(proto, steps_per_epoch=10, epochs=2), loss=proto.loss_builder())
        if mode == "graph":
            step = model.make_graphed_train_step(img, mask)
            losses = [None, None] + [float(step(img, mask)[0]) for _ in range(3)]
        else:
            losses = [float(model.train_on_batch(img, mask)[0]) for _ in range(5)]
        finals.append((losses, model.P.clone(), model.S.clone()))
        del model
        torch.cuda.empty_cache()
    (l0, p0, s0), (l1, p1, s1), (l2, p2, s2) = finals
    assert all(np.isfinite(v) for v in l0) and l0[-1] < l0[0], l0
    assert l0 == l1 and torch.equal(p0, p1) and torch.equal(s0, s1)
    assert l0[2:] == l2[2:] and torch.equal(p0, p2) and torch.equal(s0, s2)


def test_seg_training_steps_cosine_schedule(device):
    S, model, oracle, params, state, img, mask = build("bn", torch.float32, device)
    proto = S.PROTOCOLS["A"]
    opt = S.build_optimizer(proto, steps_per_epoch=2, epochs=2)          # 4 decay steps
    assert [round(opt.learning_rate(s), 6) for s in (0, 2, 4, 9)] == [1e-3, 5e-4, 0.0, 0.0]
    model.compile(optimizer=opt, loss=proto.loss_builder())
    st, adam = dict(state), {}
    for step in range(3):
        want_loss, grads, _, dice, _ = oracle.loss_and_grads(params, st, img.astype(np.float64), mask.astype(np.float64), 0.4, 0.6)
        for name in params:
            mm = adam.setdefault("m/" + name, np.zeros_like(params[name]))
            vv = adam.setdefault("v/" + name, np.zeros_like(params[name]))
            ref.adam_step(params[name], grads[name], mm, vv, step + 1, lr=opt.learning_rate(step))
        loss, d, _ = model.train_on_batch(img, mask)
        assert abs(float(loss) - want_loss) < 2e-3 * want_loss and abs(float(d) - dice) < 1e-3, step
    hist = model.fit([(img, mask)] * 2, epochs=1, verbose=0, validation_data=[(img, mask)])
    assert set(hist.history) == {"loss", "dice", "iou", "val_loss", "val_dice", "val_iou"}
    assert model.evaluate([(img, mask)], return_dict=True).keys() == {"loss", "dice", "iou"}


def test_seg_builders_contract():
    from adunet_amd import seg_model as S
    m = S.build_adaptive_depth_unet(256, 64, 4)
    assert m.name == "adaptive_unet_depth4_c64" and m.count_params() == SegUNetOracle(256, 64, 4).count_params() + sum(
        int(np.prod(s)) for s in SegUNetOracle(256, 64, 4).state_shapes.values())
    assert S.build_unet(128, 1, 64, 3).name == "unet_isic_baseline"
    with pytest.raises(ValueError):
        S.build_adaptive_depth_unet(100, 64, 4)          # input_size % 2**depth != 0
    assert S.build_unet(128, 1, 32, 4).convs["conv2d"].cout == 32       # the reference's default width (unet_vinillia.py:72)
    with pytest.raises(ValueError):
        S.build_unet(128, 1, 40, 4)                       # not a multiple of the bf16 contraction granule
    m19 = S.build_unet(128, 19, 64, 4)                    # softmax head: built for inference, training entry points refuse
    assert m19.index["mask_logits/kernel"][1] == (1, 1, 64, 19)
    with pytest.raises(NotImplementedError):
        m19.compile(loss=S.PROTOCOLS["A"].loss_builder())
    with pytest.raises(ValueError):
        S.build_unet(128, 0, 64, 4)
    assert S.PROTOCOLS["B"].loss_builder().dice_weight == 1.0 and S.PROTOCOLS["B"].batch_size == 16


@pytest.mark.parametrize("kind", ["bn", "ln"])
def test_seg_fit_replays_graphs_and_matches_eager_fit(device, kind, monkeypatch):
    """fit() of the segmentation models goes through the same per-shape hipGraph replay as the SR model (BatchNorm moving
    statistics included in what the capture-only warm-up puts back): weights, state and history equal the eager fit."""
    results = []
    for eager in ("1", "0"):
        monkeypatch.setenv("ADUNET_EAGER_FIT", eager)
        S, model, oracle, params, state, img, mask = build(kind, torch.bfloat16, device)
        proto = S.PROTOCOLS["A"]
        model.compile(optimizer=S.build_optimizer(proto, steps_per_epoch=3, epochs=2), loss=proto.loss_builder())
        rng = np.random.default_rng(9)
        data = [(rng.random((3, 32, 32, 3), dtype=np.float32), (rng.random((3, 32, 32, 1)) < 0.4).astype(np.float32))
                for _ in range(3)]
        hist = model.fit(data, epochs=2, verbose=0)
        results.append((hist.history["loss"], model.P.clone(), model.S.clone(), model.optimizer.iterations))
        if eager == "0":
            assert len(model._graph_steps) == 1
    assert results[0][3] == results[1][3] == 6
    assert results[0][0] == results[1][0]
    assert torch.equal(results[0][1], results[1][1]) and torch.equal(results[0][2], results[1][2])


def test_segmentation_under_mixed_float16(device):
    """Segmenation/code/train_adaptive_unet.py:471-476: the seg trainer's --mixed_precision is mixed_float16 too.  The
    half kernels + dynamic loss scaling drive the BatchNorm model: finite losses that fall, scaler at work."""
    from adunet_amd import seg_model as S
    from adunet_amd.model import LossScaleOptimizer
    rng = np.random.default_rng(2)
    img = rng.random((4, 32, 32, 3), dtype=np.float32)
    mask = (img[..., :1] > 0.5).astype(np.float32)                  # learnable: the mask is a threshold of the red channel
    proto = S.PROTOCOLS["B"]
    model = S.build_adaptive_depth_unet(32, 64, 2, dtype=torch.float16, device=device, seed=1)
    model.compile(optimizer=S.build_optimizer(proto, steps_per_epoch=10, epochs=1), loss=proto.loss_builder())
    assert isinstance(model.optimizer, LossScaleOptimizer)
    losses = [float(model.train_on_batch(img, mask)[0]) for _ in range(12)]
    st = model.optimizer.sync()
    assert all(np.isfinite(v) for v in losses) and losses[-1] < losses[0]
    assert st["applied"] + st["skipped"] == 12 and st["applied"] >= 8 and st["loss_scale"] >= 1.0
    assert torch.isfinite(model.P).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_build_unet_softmax_head_forward(device, dtype):
    """build_unet(num_classes > 1): Conv2D(K, 1, softmax) head (Segmenation/code/unet_vinillia.py:89-90), inference only.
    The reference defines no loss for it and never passes num_classes != 1, so training entry points must refuse."""
    from adunet_amd import seg_model as S
    p, depth, k, batch = 32, 2, 5, 3
    model = S.build_unet(p, k, 32 if dtype == torch.bfloat16 else 16, depth, dtype=dtype, device=device)
    base = model.base
    oracle = SegUNetOracle(p, base, depth, "ln", "convT", num_classes=k)
    rng = np.random.default_rng(5)
    params, state = oracle.init_params(rng)
    params[oracle.head + "/kernel"] = rng.standard_normal(oracle.param_shapes[oracle.head + "/kernel"]) * 0.7
    params[oracle.head + "/bias"] = rng.standard_normal(k) * 0.3
    params = {n: v.astype(np.float32).astype(np.float64) for n, v in params.items()}
    assert list(model.index) == list(oracle.param_shapes)
    assert model.count_params() == oracle.count_params()
    model.set_weights({n: v.astype(np.float32) for n, v in params.items()})
    img = rng.random((batch, p, p, 3), dtype=np.float32)
    storage = storage_of(model, batch) if dtype == torch.bfloat16 else None
    want = oracle.forward(params, state, img.astype(np.float64), training=False, storage=storage)
    got = model(img)
    assert got.shape == (batch, p, p, k) and got.dtype == np.float32
    assert np.abs(got.sum(axis=-1) - 1.0).max() < 1e-5
    assert np.abs(got - want).max() < (2e-4 if dtype == torch.float32 else 2e-2)
    assert (got.argmax(-1) == want.argmax(-1)).mean() > (0.999 if dtype == torch.float32 else 0.97)
    proto = S.PROTOCOLS["A"]
    with pytest.raises(NotImplementedError):
        model.compile(optimizer=S.build_optimizer(proto, 10, 2), loss=proto.loss_builder())
    with pytest.raises(NotImplementedError):
        model._forward_seg(model._to_dev(img), None, training=True, keep=True)
