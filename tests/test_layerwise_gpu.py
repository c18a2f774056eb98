"""Layer-wise ("teacher-forced") parity inside the full BASELINE configurations.

End-to-end comparisons of a reduced-precision network have a noise floor that no oracle removes: one bf16 rounding
that falls the other way (fp32 vs float64 accumulation order decides it for ~1 element in 10^4) changes the inputs of
the next layer by one unit in the last place, which flips a few dozen roundings there, and after a handful of layers
most stored values differ by one ulp (tools/diag_seg.py prints the cascade).  So the end-to-end tests bound the noise,
and THIS file checks the arithmetic: the model records every tensor each step reads and writes (`Model.audit`), and
each step is recomputed in float64 by the oracle's op ON THE PRODUCT'S OWN INPUTS.  Errors cannot compound, so the
bounds are those of a single kernel:

* tensors stored in bf16: |got - want| <= 2^-8 |want| + 3e-5 max|want|  (half an ulp of rounding, one ulp if the fp32
  sum landed on the other side of a rounding boundary, plus fp32 accumulation error relative to the tensor's scale);
* tensors stored in fp32 (fp32 path, statistics, parameter gradients): 1e-4 of the tensor's max magnitude
  (parameter gradients are sums over up to 131 072 pixels accumulated in fp32);
* ReLU / clip kinks: where a pre-activation is within 1e-5 of the kink fp32 rounding decides the side, and both
  one-sided derivatives are valid: such pixels are left out of the element-wise comparison and the reductions are
  allowed the difference between the two choices.

This is the test that takes the 1x1x1024 bottleneck, split-K launches, 2-image / 16-image tiles, the wave-specialised
kernels and the fused Conv->LayerNorm->ReLU launches of K2' (and R3's 128/64/32 pyramid) to the oracle inside the
model they are benchmarked in.  Reference arithmetic: Super_resolution/code/train_adaptive_unet.py:200-287,308-334.
"""
import numpy as np
import pytest
import torch

from oracle import ops as ref

pytestmark = pytest.mark.gpu

KINK = 1e-5


def f64(t):
    # (converted on the device: the host would spend a second per 268 MB tensor in astype)
    return t.detach().to(torch.float64).cpu().numpy()


def _tt(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def stored_tol(want, store):
    """Bound of one stored element.  `store` = relative half-ulp-or-one-ulp bound of the storage type: 2^-8 (bf16), 2^-11
    (IEEE half; plus half a subnormal quantum, 2^-25, because a scaled half gradient can fall below 2^-14), None (fp32)."""
    scale = np.abs(want).max() + 1e-30
    if not store:
        return np.full_like(want, 3e-5 * scale), scale
    return store * np.abs(want) + 3e-5 * scale + (2.0 ** -25 if store < 2.0 ** -9 else 0.0), scale


def _exceeds(got, want, store, ok_pixels=None):
    """(number of elements beyond stored_tol, largest |got - want| / scale) -- the arithmetic of stored_tol on torch's
    multi-threaded CPU kernels, a few passes over the tensor instead of NumPy's dozen single-threaded temporaries (a 268 MB
    tensor took 4.5 s per comparison)."""
    g, w = _tt(got), _tt(want)
    scale = float(w.abs().max()) + 1e-30
    err = (g - w).abs_()
    floor = 3e-5 * scale + (2.0 ** -25 if (store and store < 2.0 ** -9) else 0.0)
    tol = w.abs().mul_(store).add_(floor) if store else None
    bad = err > (tol if store else floor)
    if ok_pixels is not None:
        ok = _tt(ok_pixels).unsqueeze(-1)
        bad &= ok
        err = err * ok
    return int(bad.sum()), float(err.max()) / scale


def check_stored(got_t, want, what, store):
    nbad, worst = _exceeds(f64(got_t), want, store)
    assert nbad == 0, (what, nbad, worst)


def check_f32(got, want, what, tol=1e-4, slack=None):
    got = np.asarray(got, np.float64)
    lim = tol * (np.abs(want).max() + 1e-30) + (0 if slack is None else slack)
    err = np.abs(got - want)
    assert (err <= lim).all(), (what, float((err / (np.abs(want).max() + 1e-30)).max()))


def ln_bwd_bracket(z, mean, rstd, gam, bet, d_act):
    """LayerNorm + ReLU backward of the oracle on the product's own stored z / mean / rstd, for the three ReLU thresholds
    (0, +KINK, -KINK) that bracket a pre-activation on the kink, and the pixels WITHOUT such an element.  The element-wise
    preparation (xhat, y, masks) runs on torch's CPU kernels; the backward itself is oracle.ops.layernorm_bwd."""
    zs = _tt(f64(z))
    shp = tuple(zs.shape[:-1]) + (1,)
    mu = mean.detach().to(torch.float64).cpu().reshape(shp)
    rs = rstd.detach().to(torch.float64).cpu().reshape(shp)
    xhat = (zs - mu).mul_(rs)
    y = xhat * _tt(np.asarray(gam, np.float64)) + _tt(np.asarray(bet, np.float64))
    d = _tt(d_act)
    xh_np, rs_np = xhat.numpy(), rs.numpy()
    res = [ref.layernorm_bwd((d * (y > thr)).numpy(), gam, (xh_np, rs_np)) for thr in (0.0, KINK, -KINK)]
    ok = ~((y.abs() <= KINK).any(dim=-1)).numpy()
    return res, ok


def audit_sr_step(model, lr, hr):
    """Runs forward + loss + backward with the audit on and recomputes every recorded step."""
    from adunet_amd import _lib, ops
    lib = _lib.load()
    # `bf16` below = "16-bit storage": the bound of one stored element (2^-8 bf16, 2^-11 half), falsy on the fp32 path
    bf16 = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}.get(model.dtype)
    q = {torch.bfloat16: ref.bf16_round, torch.float16: ref.fp16_round}.get(model.dtype, lambda a: a)
    W = {k: v.astype(np.float64) for k, v in model.get_weights().items()}
    model.audit = []
    out, loss, psnr, (tape, x, t) = model.forward_loss(lr, hr, keep=True)
    model._backward(tape, x, t, 1.0 / x.numel())
    torch.cuda.synchronize()
    records, model.audit = model.audit, None
    G = {k: v.astype(np.float64) for k, v in model.get_grads().items()}
    n = lr.shape[0]
    seen = set()
    pw_variants = set()       # which bank-GEMM kernels the factored up-convs of this model launch (ad_pw_gemm_variant)
    # mixed_float16 (train_adaptive_unet.py:471-477): the head's backward multiplies the loss gradient by the dynamic loss
    # scale (a power of two, so exact); every gradient behind it, parameter gradients included, carries the factor
    sc = model._scaler()
    loss_scale = float(sc.state[0]) if sc is not None else 1.0
    assert (model.dtype == torch.float16) == (sc is not None) and (sc is None or loss_scale == 2.0 ** 15)

    def conv_input(name, x1, x2):
        cin = model.convs[name].cin
        a = f64(x1)
        if x1.dtype == torch.float32 and bf16:
            a = q(a)                                           # the first-layer kernels stage the raw batch in 16 bits
        a = a[..., :cin] if x2 is None else np.concatenate([a, f64(x2)], axis=-1)
        assert a.shape[-1] == cin
        return a

    def is_fused(name, x1, x2):
        cs = model.convs[name]
        if not bf16:
            return False
        if x1.dtype == torch.float32:
            return True                                        # dedicated 3-channel kernel: conv + LayerNorm in one
        c1, c2 = x1.shape[-1], (x2.shape[-1] if x2 is not None else 0)
        return bool(lib.ad_conv3x3_ln_relu_is_fused(n, cs.hw, cs.hw, c1, c2, cs.cout, ops.dt(model.dtype)))

    for rec in records:
        kind, name = rec[0], rec[1]
        seen.add(kind)
        if kind == "fwd_cla":
            _, _, x1, x2, z, a, mean, rstd = rec
            ln = model.convs[name].ln
            want_z = ref.conv2d_same_fwd(conv_input(name, x1, x2), q(W[name + "/kernel"]), W[name + "/bias"])
            check_stored(z, want_z, name + " z", bf16)
            zin = want_z if is_fused(name, x1, x2) else f64(z)  # statistics: fp32 accumulators (one kernel) or stored z
            y, (xhat, rs) = ref.layernorm_fwd(zin, W[ln + "/gamma"], W[ln + "/beta"])
            check_f32(mean.cpu().numpy().reshape(zin.shape[:-1]), zin.mean(axis=-1), name + " mean", 1e-5,
                      slack=1e-6 * np.abs(zin).max())
            check_f32(rstd.cpu().numpy().reshape(zin.shape[:-1]), rs[..., 0], name + " rstd", 1e-4)
            check_stored(a, ref.relu_fwd(y), name + " act", bf16)
        elif kind == "fwd_resize":
            _, _, xin, xout = rec
            check_stored(xout, ref.resize_aa_fwd(f64(xin), xout.shape[1], xout.shape[2]), name + " fwd", bf16)
        elif kind == "fwd_ca":
            _, _, xin, u = rec
            want = ref.relu_fwd(ref.conv2d_same_fwd(f64(xin), q(W[name + "/kernel"]), W[name + "/bias"]))
            check_stored(u, want, name + " relu(conv)", bf16)
        elif kind == "fwd_bank":
            # factored up-conv (csrc/upconv.hip), step 1: nine 1x1 convolutions on the low-resolution map
            _, _, xin, yb = rec
            m_, k_ = xin.numel() // xin.shape[-1], xin.shape[-1]
            pw_variants.add(lib.ad_pw_gemm_variant(m_, k_, yb.shape[-1], ops.dt(model.dtype)))          # Y = x Bank
            pw_variants.add(lib.ad_pw_gemm_variant(m_, yb.shape[-1], k_, ops.dt(model.dtype)))          # dx = dY Bank^T (bwd_caf)
            want = ref.upconv_bank_fwd(f64(xin), q(W[name + "/kernel"]))
            check_stored(yb, want.reshape(yb.shape), name + " 1x1 bank", bf16)
        elif kind == "fwd_gather":
            # ... step 2: out = relu(b + sum_tap (U Y_tap)[p + tap]) on the product's own bank
            _, _, yb, u = rec
            seen.add("fwd_ca")
            nb, hb, wb, c9 = yb.shape
            want = ref.relu_fwd(ref.upconv_gather_fwd(f64(yb).reshape(nb, hb, wb, 9, c9 // 9), W[name + "/bias"], u.shape[1], u.shape[2]))
            check_stored(u, want, name + " relu(gather)", bf16)
        elif kind == "bwd_caf":
            _, _, xin, u, d_in, dz, dyb, d = rec
            seen.add("bwd_ca")
            if dz is not d_in:            # (the same tensor when the dgrad above already applied this ReLU's gradient)
                want_dz = f64(d_in) * (f64(u) > 0)
                assert np.array_equal(f64(dz), want_dz), name + " relu grad"
                check_f32(G[name + "/bias"], want_dz.reshape(-1, want_dz.shape[-1]).sum(0), name + "/bias grad")
            nb, hb, wb, cin = xin.shape
            want_dyb = ref.upconv_gather_bwd(f64(dz), hb, wb)
            check_stored(dyb, want_dyb.reshape(dyb.shape), name + " gather^T", bf16)
            dx, dw = ref.upconv_bank_bwd(f64(xin), q(W[name + "/kernel"]), f64(dyb).reshape(want_dyb.shape))
            check_f32(G[name + "/kernel"], dw, name + "/kernel grad")
            check_stored(d, dx, name + " dx (1x1 bank)", bf16)
        elif kind == "fwd_head":
            _, _, xh, inp, target, o, stats = rec
            r = ref.conv2d_same_fwd(f64(xh), W["residual_rgb/kernel"], W["residual_rgb/bias"])
            want_out, pre = ref.clip_add_fwd(f64(inp), r)
            check_f32(f64(o), want_out, "enhanced_rgb", 1e-5)
            st = stats.cpu().numpy()
            assert abs(st[2] - ref.charbonnier_fwd(f64(target), want_out)) < 1e-5 * st[2]
            assert abs(st[1] - np.mean(ref.psnr_per_image(f64(target), want_out))) < 1e-3          # dB
        elif kind == "bwd_head":
            _, _, xh, inp, target, gscale, d = rec
            r = ref.conv2d_same_fwd(f64(xh), W["residual_rgb/kernel"], W["residual_rgb/bias"])
            want_out, pre = ref.clip_add_fwd(f64(inp), r)
            assert abs(gscale * want_out.size - 1.0) < 1e-12
            dout = ref.charbonnier_bwd(f64(target), want_out) * loss_scale
            dr = ref.clip_add_bwd(dout, pre)
            edge = (np.abs(pre) < KINK) | (np.abs(pre - 1.0) < KINK)          # clip kinks: either side is valid
            dxh, dw, db = ref.conv2d_same_bwd(f64(xh), W["residual_rgb/kernel"], dr)
            ok = ~edge.any(axis=-1)
            got = f64(d)
            check_stored_masked(got, dxh, ok, "d head activations", bf16)
            slack = np.abs(dout * edge).sum() * np.abs(f64(xh)).max()
            check_f32(G["residual_rgb/kernel"], dw, "residual_rgb/kernel grad", slack=slack)
            check_f32(G["residual_rgb/bias"], db, "residual_rgb/bias grad", slack=slack)
        elif kind == "bwd_head_ln":
            # head backward + LayerNorm/ReLU backward of the layer feeding the head in one kernel: the gradient of the
            # head activations is never stored, so the oracle chains the two steps without rounding in between
            _, _, xh, inp, target, gscale, cname, z, mean, rstd, dz = rec
            seen.add("bwd_head")
            r = ref.conv2d_same_fwd(f64(xh), W["residual_rgb/kernel"], W["residual_rgb/bias"])
            want_out, pre = ref.clip_add_fwd(f64(inp), r)
            dout = ref.charbonnier_bwd(f64(target), want_out) * loss_scale
            dr = ref.clip_add_bwd(dout, pre)
            edge = (np.abs(pre) < KINK) | (np.abs(pre - 1.0) < KINK)
            dxh, dw, db = ref.conv2d_same_bwd(f64(xh), W["residual_rgb/kernel"], dr)
            slack = np.abs(dout * edge).sum() * np.abs(f64(xh)).max()
            check_f32(G["residual_rgb/kernel"], dw, "residual_rgb/kernel grad", slack=slack)
            check_f32(G["residual_rgb/bias"], db, "residual_rgb/bias grad", slack=slack)
            ln = model.convs[cname].ln
            gam, bet = W[ln + "/gamma"], W[ln + "/beta"]
            res, ok = ln_bwd_bracket(z, mean, rstd, gam, bet, dxh)
            ok = ok & ~edge.any(axis=-1)
            check_stored_masked(f64(dz), res[0][0], ok, cname + " dz (fused with the head)", bf16)
            for j, pname in ((1, ln + "/gamma"), (2, ln + "/beta")):
                check_f32(G[pname], res[0][j], pname + " grad", slack=np.abs(res[1][j] - res[2][j]) + slack)
        elif kind == "bwd_cla":
            _, _, x1, x2, z, mean, rstd, d_in, dz, d, dsk, fused_relu = rec
            ln = model.convs[name].ln
            if d_in is not None:              # (None: this layer's dz came out of the fused head kernel, checked above)
                gam, bet = W[ln + "/gamma"], W[ln + "/beta"]
                res, ok = ln_bwd_bracket(z, mean, rstd, gam, bet, f64(d_in))
                check_stored_masked(f64(dz), res[0][0], ok, name + " dz", bf16)
                for j, pname in ((1, ln + "/gamma"), (2, ln + "/beta")):
                    check_f32(G[pname], res[0][j], pname + " grad", slack=np.abs(res[1][j] - res[2][j]))
            # conv backward on the product's own dz
            dzp = f64(dz)
            xin = conv_input(name, x1, x2)
            need_dx = d is not None
            dx, dw, db = ref.conv2d_same_bwd(xin, q(W[name + "/kernel"]), dzp, need_dx=need_dx)
            check_f32(G[name + "/kernel"], dw, name + "/kernel grad")
            check_f32(G[name + "/bias"], db, name + "/bias grad")
            if need_dx:
                c1 = d.shape[-1] if dsk is not None else model.convs[name].cin
                want_d = dx[..., :c1]
                if fused_relu:        # the dgrad kernel also applied the up-conv's ReLU gradient (x1 = its ReLU output)
                    want_d = want_d * (f64(x1) > 0)
                    seen.add("fused_relu_grad")
                check_stored(d[..., :c1] if dsk is None else d, want_d, name + " dgrad", bf16)
                if dsk is not None:
                    check_stored(dsk, dx[..., c1:], name + " dgrad (skip half)", bf16)
        elif kind == "bwd_dgrad_ln":
            # dgrad of conv `name` + LayerNorm/ReLU backward of the layer below it (whose activation was this conv's input)
            # in one kernel: the oracle chains the two steps without rounding the activation gradient in between
            _, _, dz_up, cname, z, mean, rstd, dz = rec
            da, _, _ = ref.conv2d_same_bwd(np.zeros(z.shape), q(W[name + "/kernel"]), f64(dz_up))
            ln = model.convs[cname].ln
            gam, bet = W[ln + "/gamma"], W[ln + "/beta"]
            res, ok = ln_bwd_bracket(z, mean, rstd, gam, bet, da)
            check_stored_masked(f64(dz), res[0][0], ok, cname + " dz (fused with the dgrad of " + name + ")", bf16)
            for j, pname in ((1, ln + "/gamma"), (2, ln + "/beta")):
                check_f32(G[pname], res[0][j], pname + " grad", slack=np.abs(res[1][j] - res[2][j]))
        elif kind == "bwd_ca":
            _, _, xin, u, d_in, dz, d = rec
            want_dz = f64(d_in) * (f64(u) > 0)
            assert np.array_equal(f64(dz), want_dz), name + " relu grad"
            dx, dw, db = ref.conv2d_same_bwd(f64(xin), q(W[name + "/kernel"]), want_dz)
            check_f32(G[name + "/kernel"], dw, name + "/kernel grad")
            check_f32(G[name + "/bias"], db, name + "/bias grad")
            check_stored(d, dx, name + " dgrad", bf16)
        elif kind == "bwd_resize_ln":
            # skip-gradient junction + LayerNorm/ReLU backward of the block that produced the skip, one kernel: the
            # oracle chains resize^T, the add and the LayerNorm backward without rounding in between
            _, _, d_in, dskip, cname, z, mean, rstd, dz = rec
            seen.add("bwd_resize")
            da = ref.resize_aa_bwd(f64(d_in), dskip.shape[1], dskip.shape[2]) + f64(dskip)
            ln = model.convs[cname].ln
            gam, bet = W[ln + "/gamma"], W[ln + "/beta"]
            res, ok = ln_bwd_bracket(z, mean, rstd, gam, bet, da)
            check_stored_masked(f64(dz), res[0][0], ok, cname + " dz (fused with the skip junction)", bf16)
            for j, pname in ((1, ln + "/gamma"), (2, ln + "/beta")):
                check_f32(G[pname], res[0][j], pname + " grad", slack=np.abs(res[1][j] - res[2][j]))
        elif kind == "bwd_resize":
            _, _, d_in, before, d = rec
            want = ref.resize_aa_bwd(f64(d_in), d.shape[1], d.shape[2])
            if before is not None:
                want = want + f64(before)
            check_stored(d, want, name + " bwd", bf16)
    extra = {"fused_relu_grad", "bwd_head_ln", "bwd_resize_ln", "bwd_dgrad_ln", "fwd_bank", "fwd_gather", "bwd_caf"}
    assert seen - extra == {"fwd_cla", "fwd_resize", "fwd_ca", "fwd_head", "bwd_head", "bwd_cla", "bwd_ca", "bwd_resize"}
    # (a fused dgrad + LayerNorm backward adds a record of its own next to the two "bwd_cla" records it spans; a factored
    # up-conv's backward is one record where the resize + conv pair has two)
    nrec = sum(r[0] != "bwd_dgrad_ln" for r in records) + sum(r[0] == "bwd_caf" for r in records)
    return nrec, ("fused_relu_grad" in seen, "bwd_dgrad_ln" in seen), sum(r[0] == "bwd_caf" for r in records), pw_variants


def check_stored_masked(got, want, ok_pixels, what, store):
    nbad, worst = _exceeds(got, want, store, ok_pixels)
    assert nbad == 0, (what, nbad, worst)
    # a pixel is left out when ANY of its C pre-activations lies within KINK of zero: the expected share grows with C
    # (observed 1.8 % at 1 024 channels, the 0.7 / depth 5 pyramid), so the allowance does too
    allowed = max(8, 0.01 * ok_pixels.size * max(1.0, want.shape[-1] / 256.0))
    assert (~ok_pixels).sum() <= allowed, (what, "too many pixels on a kink", int((~ok_pixels).sum()))


def build(scale, depth, p, dtype, device):
    from adunet_amd.model import Adam, build_losses_and_metrics, build_super_resolution_unet
    model, _ = build_super_resolution_unet(scale, depth_override=depth, input_size=p, dtype=dtype, device=device)
    loss, metrics = build_losses_and_metrics("charbonnier")
    model.compile(optimizer=Adam(1e-3), loss=loss, metrics=metrics)
    model._require_device()
    model.set_weights(model.initial_weights(np.random.default_rng(1234), head_uniform=0.05))
    return model


def synth(rng, n, p):
    hr = rng.random((n, p, p, 3), dtype=np.float32)
    lr = np.clip(hr + 0.05 * rng.standard_normal(hr.shape).astype(np.float32), 0, 1).astype(np.float32)
    return lr, hr


F32, BF16, F16 = torch.float32, torch.bfloat16, torch.float16
CONFIGS = [
    # name, scale, depth, patch, batch, dtypes
    ("small-ragged", 0.6, 3, 40, 3, (F32, BF16, F16)),
    ("K2p", 0.25, 4, 256, 2, (F32, BF16)),   # BASELINE `metric` headline / config 2 (depth 4, x4): pyramid 256/64/16/4/1
    ("R3", 0.5, 3, 256, 1, (F32, BF16)),     # the reference's own Experiment-1 shape: 256/128/64/32
    # batch 8: enough tiles for the wave-specialised launches incl. the fused ReLU-grad / LayerNorm-backward dgrads; fp16 is
    # the reference's GPU policy (train_adaptive_unet.py:471-477).  (fp32 runs the generic kernels at any batch: see K2p.)
    ("K2p-b8", 0.25, 4, 256, 8, (BF16, F16)),
    # BASELINE config 2 as written (P512): pyramid 512/128/32/8/2 -- the 8 x 8 and 2 x 2 levels take other small-map routes than
    # K2p's 4 x 4 / 1 x 1; two images of 512^2 are as many pixels as K2p-b8
    ("K2-b2", 0.25, 4, 512, 2, (BF16,)),
    # the reference's own Experiment-2 shapes (Super_resolution/sbatch_scripts/run_experiment_adaptive_depth.sh:36-66,
    # depth table :47-55): fractional pyramids through the wave-specialised kernels, odd widths in the skip junctions
    # fp16 is what the reference ran these rows in (train_adaptive_simple.sbatch default --mixed_precision)
    ("E2s06-b8", 0.6, 4, 256, 8, (BF16, F16)),   # 256/154/93/56/34, 64..1024 channels
    ("E2s06-b2", 0.6, 4, 256, 2, (F32,)),
    ("E2s07-b2", 0.7, 5, 256, 2, (BF16, F16)),   # 256/180/126/89/63/45, 2048-channel bottleneck (138 M parameters)
    ("E2s07-b1", 0.7, 5, 256, 1, (F32,)),        # (fp32 runs the generic kernels at any batch: one image halves the oracle's work)
    # scale 0.8 / depth 5 (the last row of the Experiment-2 table): 256/205/164/132/106/85.  Its 132 -> 164 up-conv at 256 channels
    # stages 2 x 32 256 bytes per workgroup in the forward gather: the eight-slot variant BEYOND the 64 KB default dynamic-LDS limit
    # (until r04 only the 12-slot instantiations had the limit raised, so this row's training would have aborted)
    ("E2s08-b1", 0.8, 5, 256, 1, (BF16,)),
]
BIG_LAUNCH_CONFIGS = {"K2p-b8", "E2s06-b8", "K2-b2"}              # batch 8: >= 1 work item per CU at full resolution
CASES = [(c, dt_) for c in CONFIGS for dt_ in c[5]]


@pytest.mark.parametrize("cfg,dtype", CASES, ids=[f"{c[0]}-{str(d).split('.')[-1]}" for c, d in CASES])
def test_every_step_of_the_model_against_the_oracle(device, cfg, dtype):
    _, scale, depth, p, n, _ = cfg
    model = build(scale, depth, p, dtype, device)
    lr, hr = synth(np.random.default_rng(4321), n, p)
    nrec, (fused_relu, fused_ln), nfactored, pw_variants = audit_sr_step(model, lr, hr)
    # decoder levels whose source map is at least 16 pixels wide run the up-conv in the factored form, in every dtype --
    # except where a workgroup's piece of a bank row exceeds the forward gather's staging window (fp32 from 512 output
    # channels on, 16-bit from 1 024: the deepest levels of the 0.6 / 0.7 pyramids), which keep the resize + 3x3 pair
    assert nfactored == len(model._factored_upconvs()), (nfactored, model._factored_upconvs())
    wide = 512 if dtype == torch.float32 else 1024
    assert nfactored == sum(sz >= 16 and cs.cout < wide for sz, cs in
                            zip(model.sizes[1:], [st[1] for st in reversed(model._plan) if st[0] == "upconv"])), (nfactored, model.sizes)
    if cfg[0] in ("E2s06-b8", "E2s07-b2") and dtype != torch.float32:
        # the deep Experiment-2 levels (K = 256 ... 4 608) are the only audited shapes whose bank GEMMs take the LDS-tiled kernel
        # with TWO k-stages of loads in flight (K2p's GEMMs have K = 128 / 576: one stage)
        assert 2 in pw_variants and 1 in pw_variants, pw_variants
    if cfg[0] in BIG_LAUNCH_CONFIGS:  # the weights-resident kernels with the fused ReLU-grad / LayerNorm-backward epilogues
        assert fused_relu and fused_ln, (fused_relu, fused_ln)
    # forward: 2 convs per block (2 depth + 2 blocks), depth up-convs, 2 depth resizes, head; backward: the same again
    assert nrec == 2 * (2 * (2 * depth + 2) + depth + 2 * depth + 1)


def test_audit_is_off_by_default_and_does_not_change_results(device):
    model = build(0.5, 2, 32, torch.bfloat16, device)
    lr, hr = synth(np.random.default_rng(1), 2, 32)
    assert model.audit is None
    out0, loss0, _, (tape, x, t) = model.forward_loss(lr, hr, keep=True)
    model._backward(tape, x, t, 1.0 / x.numel())
    g0 = model.G.clone()
    model.audit = []
    out1, loss1, _, (tape, x, t) = model.forward_loss(lr, hr, keep=True)
    model._backward(tape, x, t, 1.0 / x.numel())
    assert torch.equal(out0, out1) and torch.equal(g0, model.G) and len(model.audit) > 10


# ----------------------------------------------------------------------------- segmentation models (tier 2, K3)
def audit_seg_step(S, model, img, mask, bce_w=0.4, dice_w=0.6):
    """Same idea for build_adaptive_depth_unet / build_unet (Segmenation/code/train_adaptive_unet.py:325-362,
    unet_vinillia.py:42-91): BatchNorm statistics and backward, MaxPool, bilinear x2, Conv2DTranspose, sigmoid head."""
    from adunet_amd import _lib, ops
    lib = _lib.load()
    bf16 = 2.0 ** -8 if model.dtype == torch.bfloat16 else None      # bound of one stored element (see stored_tol)
    q = ref.bf16_round if bf16 else (lambda a: a)
    W = {k: v.astype(np.float64) for k, v in model.get_weights().items()}
    model.audit = []
    x, m = model._to_dev(img), model._to_dev_mask(mask)
    prob, sums, tape = model._forward_seg(x, m, training=True, keep=True)
    model._backward_seg(tape, m)
    torch.cuda.synchronize()
    records, model.audit = model.audit, None
    G = {k: v.astype(np.float64) for k, v in model.get_grads().items()}
    n = img.shape[0]
    bn = model.norm == "bn"
    seen = set()

    def conv_input(name, x1, x2):
        cin = model.convs[name].cin
        a = f64(x1)
        if x1.dtype == torch.float32 and bf16:
            a = q(a)                                           # the first-layer kernels stage the raw image in 16 bits
        a = a[..., :cin] if x2 is None else np.concatenate([a, f64(x2)], axis=-1)
        assert a.shape[-1] == cin
        return a

    for rec in records:
        kind, name = rec[0], rec[1]
        seen.add(kind)
        if kind == "fwd_cna":
            _, _, x1, x2, z, a, mean, rstd, training = rec
            nn = model.convs[name].ln
            want_z = ref.conv2d_same_fwd(conv_input(name, x1, x2), q(W[name + "/kernel"]), W[name + "/bias"])
            check_stored(z, want_z, name + " z", bf16)
            if bn:
                y, (xhat, rs), mu, var = ref.batchnorm_train_fwd(f64(z), W[nn + "/gamma"], W[nn + "/beta"])
                check_f32(mean.cpu().numpy(), mu, name + " batch mean", 1e-5, slack=1e-6 * np.abs(f64(z)).max())
                check_f32(rstd.cpu().numpy(), rs, name + " batch rstd", 1e-4)
            else:
                c1, c2 = x1.shape[-1], (x2.shape[-1] if x2 is not None else 0)
                fused = bool(bf16) and (x1.dtype == torch.float32 or       # (raw image: the 3-channel kernel, conv + LayerNorm in one)
                                        bool(lib.ad_conv3x3_ln_relu_is_fused(n, z.shape[1], z.shape[2], c1, c2, z.shape[3], ops.dt(model.dtype))))
                y, _ = ref.layernorm_fwd(want_z if fused else f64(z), W[nn + "/gamma"], W[nn + "/beta"])
            check_stored(a, ref.relu_fwd(y), name + " act", bf16)
        elif kind == "fwd_pool":
            assert np.array_equal(f64(rec[3]), ref.maxpool2_fwd(f64(rec[2]))), name
        elif kind == "fwd_up2":
            check_stored(rec[3], ref.upsample2_bilinear_fwd(f64(rec[2])), name + " bilinear x2", bf16)
        elif kind == "fwd_convT":
            want = ref.conv_transpose2x2s2_fwd(f64(rec[2]), q(W[name + "/kernel"]), W[name + "/bias"])
            check_stored(rec[3], want, name + " fwd", bf16)
        elif kind == "fwd_head":
            _, _, xh, msk, p, sm = rec
            logit = ref.conv2d_same_fwd(f64(xh), W[name + "/kernel"], W[name + "/bias"])
            check_f32(f64(p), ref.sigmoid(logit), "probabilities", 1e-5)
            loss, dice, iou = model._metrics_from(sm, float(msk.numel()))
            want_loss, _ = ref.seg_loss_fwd_bwd(f64(msk), ref.sigmoid(logit), bce_w, dice_w)
            assert abs(float(loss) - want_loss) < 1e-4 * want_loss
            assert abs(float(dice) - ref.dice_coefficient(f64(msk), ref.sigmoid(logit))) < 1e-5
            assert abs(float(iou) - ref.iou_score(f64(msk), ref.sigmoid(logit))) < 1e-5
        elif kind == "bwd_head":
            _, _, xh, msk, p, d = rec
            pp = f64(p)                                             # the product's own probabilities
            _, dp = ref.seg_loss_fwd_bwd(f64(msk), pp, bce_w, dice_w)
            dxh, dw, db = ref.conv2d_same_bwd(f64(xh), W[name + "/kernel"], dp * pp * (1 - pp))
            check_stored(d, dxh, "d head activations", bf16)
            check_f32(G[name + "/kernel"], dw, name + "/kernel grad")
            check_f32(G[name + "/bias"], db, name + "/bias grad")
        elif kind == "bwd_cna":
            _, _, x1, x2, z, mean, rstd, d_in, dz, d, dsk = rec
            nn = model.convs[name].ln
            gam, bet = W[nn + "/gamma"], W[nn + "/beta"]
            zs = f64(z)
            shp = (1, 1, 1, -1) if bn else zs.shape[:-1] + (1,)
            mu = mean.cpu().numpy().astype(np.float64).reshape(shp)
            rs = rstd.cpu().numpy().astype(np.float64).reshape(shp)
            xhat = (zs - mu) * rs
            y = xhat * gam + bet
            din = f64(d_in)
            bwd = ref.batchnorm_train_bwd if bn else ref.layernorm_bwd
            cache = (xhat, rs.reshape(-1) if bn else rs)
            res = [bwd(din * (y > thr), gam, cache) for thr in (0.0, KINK, -KINK)]
            kink = np.abs(y) <= KINK
            if bn:       # a flipped element moves its whole channel's batch sums: allow the bracket on the affected channels
                slack_dz = np.abs(res[1][0] - res[2][0]).max(axis=(0, 1, 2), keepdims=True) * 2
                ok = ~kink
                got, want = f64(dz), res[0][0]
                scale = np.abs(want).max() + 1e-30
                tol = (2.0 ** -8 * np.abs(want) + 3e-5 * scale if bf16 else 3e-5 * scale) + slack_dz
                assert not ((np.abs(got - want) > tol) & ok).any(), name + " dz"
            else:
                check_stored_masked(f64(dz), res[0][0], ~kink.any(axis=-1), name + " dz", bf16)
            for j, pname in ((1, nn + "/gamma"), (2, nn + "/beta")):
                check_f32(G[pname], res[0][j], pname + " grad", slack=np.abs(res[1][j] - res[2][j]))
            dzp = f64(dz)
            need_dx = d is not None
            dx, dw, db = ref.conv2d_same_bwd(conv_input(name, x1, x2), q(W[name + "/kernel"]), dzp, need_dx=need_dx)
            check_f32(G[name + "/kernel"], dw, name + "/kernel grad")
            # behind BatchNorm sum(dz) is zero in exact arithmetic: allow the fp32 summation error of the terms
            check_f32(G[name + "/bias"], db, name + "/bias grad", slack=2e-6 * np.abs(dzp).sum(axis=(0, 1, 2)))
            if need_dx:
                c1 = d.shape[-1] if dsk is not None else model.convs[name].cin
                check_stored(d[..., :c1] if dsk is None else d, dx[..., :c1], name + " dgrad", bf16)
                if dsk is not None:
                    check_stored(dsk, dx[..., c1:], name + " dgrad (skip half)", bf16)
        elif kind == "bwd_up2":
            _, _, d_in, d = rec
            check_stored(d, ref.resize_aa_bwd(f64(d_in), d.shape[1], d.shape[2]), "bilinear x2 bwd", bf16)
        elif kind == "bwd_convT":
            _, _, xin, d_in, d = rec
            dx, dw, db = ref.conv_transpose2x2s2_bwd(f64(xin), q(W[name + "/kernel"]), f64(d_in))
            check_stored(d, dx, name + " dgrad", bf16)
            check_f32(G[name + "/kernel"], dw, name + "/kernel grad")
            check_f32(G[name + "/bias"], db, name + "/bias grad")
        elif kind == "bwd_pool":
            _, _, xin, d_in, skip_grad, d = rec
            check_stored(d, ref.maxpool2_bwd(f64(d_in), f64(xin)) + f64(skip_grad), name + " bwd", bf16)
    want_kinds = {"fwd_cna", "fwd_pool", "fwd_head", "bwd_head", "bwd_cna", "bwd_pool"} | (
        {"fwd_up2", "bwd_up2"} if model.up == "bilinear" else {"fwd_convT", "bwd_convT"})
    assert seen == want_kinds, seen ^ want_kinds


SEG_CONFIGS = [
    # name, builder kind, input size, depth, batch
    ("bn-small", "bn", 32, 2, 3),
    ("ln-convT-small", "ln", 32, 2, 3),
    ("ln-convT-c32", "ln32", 64, 3, 2),  # build_unet at the reference's DEFAULT base_channels = 32 (unet_vinillia.py:72)
    ("K3", "bn", 256, 5, 2),          # BASELINE config 3 as the reference expresses it: build_adaptive_depth_unet(256, 64, 5)
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("cfg", SEG_CONFIGS, ids=[c[0] for c in SEG_CONFIGS])
def test_every_step_of_the_segmentation_models_against_the_oracle(device, cfg, dtype):
    from adunet_amd import seg_model as S
    _, kind, p, depth, batch = cfg
    model = (S.build_adaptive_depth_unet(p, 64, depth, dtype=dtype, device=device, seed=5) if kind == "bn"
             else S.build_unet(p, 1, 32 if kind == "ln32" else 64, depth, dtype=dtype, device=device, seed=5))
    proto = S.PROTOCOLS["A"]
    model.compile(optimizer=S.build_optimizer(proto, steps_per_epoch=10, epochs=2), loss=proto.loss_builder())
    model._require_device()
    rng = np.random.default_rng(11)
    w = model.get_weights()
    for k in w:                                   # non-trivial gamma / beta / biases so every term of the backward is live
        if k.endswith("/gamma"):
            w[k] = rng.uniform(0.8, 1.2, w[k].shape).astype(np.float32)
        elif k.endswith("/beta") or k.endswith("/bias"):
            w[k] = rng.uniform(-0.1, 0.1, w[k].shape).astype(np.float32)
    model.set_weights(w)
    img = rng.random((batch, p, p, 3), dtype=np.float32)
    mask = (rng.random((batch, p, p, 1)) < 0.35).astype(np.float32)
    audit_seg_step(S, model, img, mask)
