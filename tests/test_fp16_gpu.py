"""The reference's own reduced-precision policy: mixed_float16 + Keras dynamic loss scaling
(Super_resolution/code/train_adaptive_unet.py:471-477; BASELINE.json config 5's precision).

The kernels are the bf16 kernels instantiated for IEEE half (v_mfma_f32_16x16x32_f16), so the parity statement is the
same: against the oracle's fp16-STORAGE mode (float64 sums, tensors rounded to half where the product stores them,
overflow -> inf as the hardware conversion does), plus the scaler's decisions, which must agree step for step.
"""
import numpy as np
import pytest
import torch

from oracle import ops as ref
from oracle.loss_scale import DynamicLossScale
from oracle.sr_unet import SRUNetOracle, Storage


def test_dynamic_loss_scale_rules():
    """CPU: the published rules (halve + skip on non-finite, double after N finite steps, floor at 1)."""
    s = DynamicLossScale(2.0 ** 15, dynamic_growth_steps=3)
    good, bad = {"g": np.ones(4)}, {"g": np.array([1.0, np.inf])}
    assert s.update(good) and s.update(good) and s.scale == 2.0 ** 15
    assert s.update(good) and s.scale == 2.0 ** 16 and s.good_steps == 0
    assert not s.update(bad) and s.scale == 2.0 ** 15 and s.applied == 3 and s.skipped == 1
    assert not s.update({"g": np.array([np.nan])}) and s.scale == 2.0 ** 14
    t = DynamicLossScale(1.5, 2)
    assert not t.update(bad) and t.scale == 1.0 and not t.update(bad) and t.scale == 1.0


def synth(rng, n, p):
    hr = rng.random((n, p, p, 3), dtype=np.float32)
    lr = np.clip(hr + 0.05 * rng.standard_normal(hr.shape).astype(np.float32), 0, 1).astype(np.float32)
    return lr, hr


def fp16_storage(model, n):
    from adunet_amd import _lib, ops
    lib = _lib.load()
    fused = {}
    first = next(iter(model.convs.values()))
    for step in model._plan:
        if step[0] != "block":
            continue
        for i, cs in enumerate(step[1]):
            c1, c2 = (cs.cin // 2, cs.cin // 2) if (step[2] is not None and i == 0) else (model._cin_pad(cs), 0)
            if cs is first and cs.cin == 3 and lib.ad_conv3x3_c3_supported(n, cs.hw, cs.hw, cs.cout, ops.dt(model.dtype)):
                fused[cs.name] = True
            else:
                fused[cs.name] = bool(lib.ad_conv3x3_ln_relu_is_fused(n, cs.hw, cs.hw, c1, c2, cs.cout, ops.dt(model.dtype)))
    return Storage(ref.fp16_round, lambda conv, *shape: fused[conv], factored=lambda conv: conv in model._factored_upconvs())


def rel(got, want):
    return float(np.abs(np.asarray(got, np.float64) - want).max() / (np.abs(want).max() + 1e-30))


def build(device, scale=0.5, depth=2, p=32, optimizer=None, head_uniform=0.05):
    from adunet_amd.model import Adam, build_losses_and_metrics, build_super_resolution_unet
    rng = np.random.default_rng(1234)
    oracle = SRUNetOracle(scale, depth, p)
    params = {k: v.astype(np.float32).astype(np.float64) for k, v in oracle.init_params(rng, head_uniform=head_uniform).items()}
    model, _ = build_super_resolution_unet(scale, depth_override=depth, input_size=p, dtype=torch.float16, device=device)
    model.set_weights({k: v.astype(np.float32) for k, v in params.items()})
    loss, metrics = build_losses_and_metrics("charbonnier")
    model.compile(optimizer=optimizer or Adam(1e-3), loss=loss, metrics=metrics)
    return oracle, params, model, rng


@pytest.mark.gpu
def test_compile_wraps_the_optimizer_like_keras(device):
    from adunet_amd.model import Adam, LossScaleOptimizer, build_losses_and_metrics, build_super_resolution_unet
    _, _, model, _ = build(device)
    assert isinstance(model.optimizer, LossScaleOptimizer) and model.optimizer.initial_scale == 2.0 ** 15
    assert model.optimizer.dynamic_growth_steps == 2000
    m32, _ = build_super_resolution_unet(0.5, depth_override=1, input_size=16, dtype=torch.float32, device=device)
    loss, metrics = build_losses_and_metrics("l1")
    m32.compile(optimizer=Adam(1e-3), loss=loss, metrics=metrics)
    assert isinstance(m32.optimizer, Adam)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(0.5, 2, 32, 2), (0.6, 3, 40, 3), (0.25, 2, 64, 2)])
def test_fp16_forward_and_gradients_against_the_fp16_storage_oracle(device, case):
    scale, depth, p, n = case
    oracle, params, model, rng = build(device, scale, depth, p)
    lr, hr = synth(rng, n, p)
    sc = 2.0 ** 15
    # the oracle differentiates the SCALED loss too: half's range (6e-8 ... 65504) is part of what is being compared
    want_out = oracle.forward(params, lr.astype(np.float64), storage=fp16_storage(model, n))
    want_loss = ref.charbonnier_fwd(hr.astype(np.float64), want_out)
    want_psnr = float(np.mean(ref.psnr_per_image(hr.astype(np.float64), want_out)))
    want_grads = {k: v / sc for k, v in oracle.backward(params, ref.charbonnier_bwd(hr.astype(np.float64), want_out) * sc).items()}
    out, loss, psnr, (tape, x, t) = model.forward_loss(lr, hr, keep=True)
    model._backward(tape, x, t, 1.0 / x.numel())                 # scaled by 2**15 in the head kernel
    assert model.optimizer.sync()["loss_scale"] == sc
    assert rel(out.cpu().numpy(), want_out) < 2e-3               # half carries 11 significant bits (bf16: 8 -> 1e-2)
    assert abs(float(loss) - want_loss) < 1e-3 * want_loss
    assert abs(float(psnr) - want_psnr) < 5e-3                   # dB
    grads = {k: v / sc for k, v in model.get_grads().items()}
    assert all(np.isfinite(v).all() for v in grads.values())
    # 1e-2 per tensor for layers that sum over >= 1024 pixels, scaled by sqrt(1024 / pixels) below that (rounding noise
    # in a sum over P pixels falls like 1/sqrt(P); same policy as tests/test_model_gpu.py)
    pixels = {}
    for cs in model.convs.values():
        for pname in (cs.name + "/kernel", cs.name + "/bias") + ((cs.ln + "/gamma", cs.ln + "/beta") if cs.ln else ()):
            pixels[pname] = n * cs.hw * cs.hw
    worst = max((rel(grads[k], want_grads[k]) / (1e-2 * max(1.0, (1024.0 / pixels[k]) ** 0.5)), k) for k in want_grads)
    assert worst[0] < 1.0, worst
    ga = np.concatenate([grads[k].reshape(-1) for k in want_grads]).astype(np.float64)
    gb = np.concatenate([want_grads[k].reshape(-1) for k in want_grads])
    assert float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb))) > 0.99999


@pytest.mark.gpu
def test_loss_scale_trajectory_with_forced_overflow(device):
    """Start far too high (2**30: the gradients of the head block overflow half): the first steps must be skipped with the
    scale halving each time, then training proceeds and the scale doubles every 3 finite steps -- decisions, scale and
    iteration count equal to the oracle's restatement step for step; weights untouched by skipped steps."""
    from adunet_amd.model import Adam, LossScaleOptimizer
    opt = LossScaleOptimizer(Adam(1e-3), initial_scale=2.0 ** 30, dynamic_growth_steps=3)
    oracle, params, model, rng = build(device, optimizer=opt)
    storage = fp16_storage(model, 4)
    scaler = DynamicLossScale(2.0 ** 30, 3)
    state = {}
    history = []
    for step in range(14):
        lr, hr = synth(rng, 4, 32)
        before = model.P.clone()
        want_loss, want_psnr = oracle.train_step(params, state, lr.astype(np.float64), hr.astype(np.float64), lr=1e-3,
                                                 storage=storage, scaler=scaler)
        loss, psnr = model.train_on_batch(lr, hr)
        st = model.optimizer.sync()
        history.append((st["loss_scale"], st["applied"], st["skipped"]))
        assert (st["loss_scale"], st["applied"], st["skipped"]) == (scaler.scale, scaler.applied, scaler.skipped), (step, st)
        if st["skipped"] > (history[-2][2] if len(history) > 1 else 0):
            assert torch.equal(before, model.P), "a skipped step must not touch the weights"
        else:
            assert not torch.equal(before, model.P)
        # (Adam turns a rounding difference in a tiny gradient into a full +-lr step: the trajectories drift slowly)
        assert abs(float(loss) - want_loss) < 3e-2 * want_loss, step
    assert history[0][2] == 1 and history[-1][1] >= 6            # skipped at first, trained later
    assert len({h[0] for h in history}) >= 4                      # the scale moved down and up again
    got = model.get_weights()
    worst = max((float(np.abs(got[k] - params[k]).max()), k) for k in params)
    assert worst[0] < 1e-2, worst                                 # ~10 applied Adam steps of 1e-3 each
    assert model.optimizer.iterations == scaler.applied


@pytest.mark.gpu
def test_fp16_graph_replay_equals_eager(device):
    """The scaler is device resident, so the whole scaled step (finiteness check, skip-or-apply, scale update) replays
    from a hipGraph: same trajectory as the eager step, bit for bit, through an overflow."""
    from adunet_amd.model import Adam, LossScaleOptimizer
    rng = np.random.default_rng(3)
    batches = [synth(rng, 2, 32) for _ in range(8)]
    results = []
    for graphed in (False, True):
        opt = LossScaleOptimizer(Adam(1e-3), initial_scale=2.0 ** 30, dynamic_growth_steps=2)
        _, _, model, _ = build(device, optimizer=opt)
        if graphed:
            step = model.make_graphed_train_step(*batches[0])
        else:
            step = model.train_on_batch
            step(*batches[0]); step(*batches[0])
        losses = [float(step(*b)[0]) for b in batches[1:]]
        results.append((losses, model.P.clone(), model.optimizer.sync()))
    assert results[0][0] == results[1][0] and torch.equal(results[0][1], results[1][1])
    assert results[0][2] == results[1][2] and results[0][2]["skipped"] >= 1
