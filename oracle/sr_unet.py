"""NumPy oracle of the adaptive-depth SR U-Net (test infrastructure only).

Restates ``build_super_resolution_unet`` / ``conv_block`` / the Keras train step of
/root/reference/Super_resolution/code/train_adaptive_unet.py:200-287,308-334,489-494
with explicit forward and backward passes.  Parameter names follow Keras' automatic
layer naming (conv2d, conv2d_1, layer_normalization, ..., residual_rgb) so that the
layer list can be compared line by line with the reference's ``model.summary()`` dumps.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from . import ops


def _uname(counter: Dict[str, int], base: str) -> str:
    k = counter.get(base, 0)
    counter[base] = k + 1
    return base if k == 0 else f"{base}_{k}"


class Storage:
    """Where the reduced-precision product path rounds.  `round` maps an array to the storage type's values
    (ops.bf16_round / ops.fp16_round; None = no rounding).  `fused(conv, n, h, w, cin, cout)` tells whether the product
    runs that Conv2D -> LayerNormalization link as one kernel (statistics from the fp32 accumulators) or as two
    (statistics from the stored conv output); the tests ask the library itself."""

    def __init__(self, round=None, fused=None, factored=None):
        """factored(conv) -> bool: the product runs that decoder up-conv in its factored form (nine 1x1 convolutions on the
        low-resolution map + interpolating gather, ops.upconv_*): the tensor it stores between the two kernels is the bank
        Y = x W_tap, not the up-resized activation, so that is where this mode rounds."""
        self._round = round
        self._fused = fused
        self._factored = factored

    def factored(self, conv) -> bool:
        return False if self._factored is None else bool(self._factored(conv))

    def q(self, x):
        return x if self._round is None else self._round(x)

    def fused(self, conv, n, h, w, cin, cout) -> bool:
        return False if self._fused is None else bool(self._fused(conv, n, h, w, cin, cout))


class SRUNetOracle:
    """Float64 (or float32) CPU model with manual backward."""

    def __init__(self, scale: float, depth_override: int | None = None, input_size: int = 256,
                 base_channels: int = 64, residual_head_channels: int = 64, max_depth: int = 7):
        # train_adaptive_unet.py:227-231
        self.scale = float(scale)
        self.depth = depth_override if depth_override is not None else ops.custom_depth_from_scale(
            scale, max_depth=max_depth, base_resolution=input_size)
        self.input_size = input_size
        self.base = base_channels
        self.head = residual_head_channels
        self.name = f"U-Net_SR_scale{scale:.2f}_depth{self.depth}"
        self.info = {
            "scale": scale,
            "depth": self.depth,
            "bottleneck_size": ops.estimate_bottleneck_size(input_size, scale, self.depth),
            "base_channels": base_channels,
            "max_depth": max_depth,
        }
        self.layers: List[dict] = []   # Keras summary rows: name, type, shape (H, W, C), params
        self.param_shapes: Dict[str, Tuple[int, ...]] = {}
        self._plan: List[tuple] = []
        self._build()

    # ------------------------------------------------------------------ #
    def _add(self, name, typ, shape, nparams=0):
        self.layers.append({"name": name, "type": typ, "shape": tuple(shape), "params": int(nparams)})

    def _conv(self, cnt, cin, cout, hw, k=3, name=None):
        name = name or _uname(cnt, "conv2d")
        self.param_shapes[name + "/kernel"] = (k, k, cin, cout)
        self.param_shapes[name + "/bias"] = (cout,)
        self._add(name, "Conv2D", (hw, hw, cout), k * k * cin * cout + cout)
        return name

    def _conv_block(self, cnt, cin, nf, hw):
        names = []
        for i in range(2):
            c = self._conv(cnt, cin if i == 0 else nf, nf, hw)
            ln = _uname(cnt, "layer_normalization")
            self.param_shapes[ln + "/gamma"] = (nf,)
            self.param_shapes[ln + "/beta"] = (nf,)
            self._add(ln, "LayerNormalization", (hw, hw, nf), 2 * nf)
            act = _uname(cnt, "activation")
            self._add(act, "Activation", (hw, hw, nf), 0)
            names.append((c, ln))
        return names

    def _build(self):
        cnt: Dict[str, int] = {}
        p = self.input_size
        self._add("low_res_input", "InputLayer", (p, p, 3), 0)
        nf, hw, cin = self.base, p, 3
        self.sizes = [p]
        plan = []
        enc_down_row = None
        for _ in range(self.depth):
            blk = self._conv_block(cnt, cin, nf, hw)
            plan.append(("block", blk))
            nhw = ops.resize_by_scale_size(hw, self.scale)
            plan.append(("down", hw, nhw))
            if enc_down_row is None:
                enc_down_row = len(self.layers)
                self._add("enc_down", "ResizeByScale", (nhw, nhw, nf), 0)
            else:
                self.layers[enc_down_row]["shape"] = (nhw, nhw, nf)  # shared layer: summary shows last call
            hw, cin = nhw, nf
            self.sizes.append(hw)
            nf *= 2
        blk = self._conv_block(cnt, cin, nf, hw)
        plan.append(("block", blk))
        dec_up_row = None
        for lvl in reversed(range(self.depth)):
            nf //= 2
            shw = self.sizes[lvl]
            plan.append(("up", hw, shw))
            if dec_up_row is None:
                dec_up_row = len(self.layers)
                self._add("dec_up", "ResizeToMatch", (shw, shw, 2 * nf), 0)
            else:
                self.layers[dec_up_row]["shape"] = (shw, shw, 2 * nf)
            up = self._conv(cnt, 2 * nf, nf, shw)
            plan.append(("upconv", up))
            cat = _uname(cnt, "concatenate")
            self._add(cat, "Concatenate", (shw, shw, 2 * nf), 0)
            plan.append(("concat", lvl))
            blk = self._conv_block(cnt, 2 * nf, nf, shw)
            plan.append(("block", blk))
            hw = shw
        blk = self._conv_block(cnt, nf, self.head, hw)
        plan.append(("block", blk))
        self._conv(cnt, self.head, 3, hw, k=1, name="residual_rgb")
        plan.append(("head",))
        self._add("enhanced_rgb", "ClippedResidualAdd", (hw, hw, 3), 0)
        self._plan = plan

    def count_params(self) -> int:
        return sum(int(np.prod(s)) for s in self.param_shapes.values())

    # ------------------------------------------------------------------ #
    def init_params(self, rng: np.random.Generator, dtype=np.float64, head_uniform: float = 0.0):
        """Glorot-uniform kernels, zero biases, LN gamma=1 beta=0, residual_rgb zero
        (or U(-head_uniform, head_uniform) so that gradients are non-trivial)."""
        params = {}
        for name, shape in self.param_shapes.items():
            if name.endswith("/kernel"):
                if name.startswith("residual_rgb"):
                    params[name] = (rng.uniform(-head_uniform, head_uniform, size=shape).astype(dtype)
                                    if head_uniform > 0 else np.zeros(shape, dtype))
                else:
                    params[name] = ops.glorot_uniform(rng, shape, dtype)
            elif name.endswith("/gamma"):
                params[name] = np.ones(shape, dtype)
            elif name.startswith("residual_rgb") and head_uniform > 0:
                params[name] = rng.uniform(-head_uniform, head_uniform, size=shape).astype(dtype)
            else:
                params[name] = np.zeros(shape, dtype)
        return params

    # ------------------------------------------------------------------ #
    def forward(self, params, x, keep: bool = True, storage: "Storage | None" = None):
        """storage = None: exact arithmetic in the dtype of the inputs.  storage = Storage(round, fused): the product's
        reduced-precision path, i.e. the same float64 sums with tensors rounded where the product stores them."""
        st = storage or Storage()
        q = st.q
        tape = []
        skips = []
        inp = x
        x = q(x)                                   # the first conv reads the batch in the storage type
        n = x.shape[0]
        for step in self._plan:
            kind = step[0]
            if kind == "block":
                for conv, ln in step[1]:
                    w = q(params[conv + "/kernel"])
                    z = ops.conv2d_same_fwd(x, w, params[conv + "/bias"])
                    zs = q(z)                      # the conv output as stored (read again by the backward pass)
                    # statistics: from the fp32 accumulators where conv + LayerNorm are one kernel, else from stored z
                    zin = z if st.fused(conv, n, x.shape[1], x.shape[2], x.shape[3], w.shape[3]) else zs
                    y, (xhat, rstd) = ops.layernorm_fwd(zin, params[ln + "/gamma"], params[ln + "/beta"])
                    mu = zin.mean(axis=-1, keepdims=True)
                    a = q(ops.relu_fwd(y))
                    tape.append(("cla", conv, ln, x, (zs, mu, rstd), a))
                    x = a
            elif kind == "down":
                h = x.shape[1]
                tape.append(("down", h, len(skips)))
                skips.append(x)
                x = q(ops.resize_aa_fwd(x, step[2], step[2]))
            elif kind == "up":
                pending_up = step[2]               # the resize belongs to the up-conv that follows (train_adaptive_unet.py:258-259)
            elif kind == "upconv":
                conv = step[1]
                if st.factored(conv):
                    # same function, other association (ops.upconv_*): the stored intermediate is the 1x1 bank
                    ybank = q(ops.upconv_bank_fwd(x, q(params[conv + "/kernel"])))
                    zpre = ops.upconv_gather_fwd(ybank, params[conv + "/bias"], pending_up, pending_up)
                    a = q(ops.relu_fwd(zpre))
                    tape.append(("caf", conv, x, a, zpre))
                else:
                    tape.append(("up", x.shape[1]))
                    x = q(ops.resize_aa_fwd(x, pending_up, pending_up))
                    zpre = ops.conv2d_same_fwd(x, q(params[conv + "/kernel"]), params[conv + "/bias"])
                    a = q(ops.relu_fwd(zpre))
                    tape.append(("ca", conv, x, a, zpre))
                x = a
            elif kind == "concat":
                skip = skips[step[1]]
                tape.append(("concat", x.shape[-1], step[1]))
                x = np.concatenate([x, skip], axis=-1)
            elif kind == "head":           # fp32 weights and arithmetic on the stored head activations; fp32 output
                r = ops.conv2d_same_fwd(x, params["residual_rgb/kernel"], params["residual_rgb/bias"])
                out, pre = ops.clip_add_fwd(inp, r)
                tape.append(("head", x, pre))
                x = out
        self._tape = tape if keep else None
        self._nskips = len(skips)
        self._storage = st
        return x

    def backward(self, params, dout, kink: float = 0.0):
        """kink: shifts every ReLU / clip decision by that amount (0 = the reference's masks).  Where a pre-activation
        sits within float32 rounding of a kink, a float32 implementation may land on either side and both one-sided
        derivatives are valid; backward(+d) and backward(-d) bracket what such flips can do to each gradient."""
        q = self._storage.q
        grads = {}
        dskips = [None] * self._nskips
        d = dout
        for rec in reversed(self._tape):
            kind = rec[0]
            if kind == "head":
                _, xh, pre = rec
                dr = ops.clip_add_bwd(d, pre) if kink == 0.0 else d * ((pre >= kink) & (pre <= 1.0 - kink))
                d, dw, db = ops.conv2d_same_bwd(xh, params["residual_rgb/kernel"], dr)
                d = q(d)
                grads["residual_rgb/kernel"], grads["residual_rgb/bias"] = dw, db
            elif kind == "cla":
                _, conv, ln, xin, (zs, mu, rstd), a = rec
                # LayerNorm + ReLU backward from what was saved: stored z, mean, rstd (the ReLU mask is re-derived)
                xhat = (zs - mu) * rstd
                dy = d * (xhat * params[ln + "/gamma"] + params[ln + "/beta"] > kink)
                dz, dg, dbeta = ops.layernorm_bwd(dy, params[ln + "/gamma"], (xhat, rstd))
                dz = q(dz)
                grads[ln + "/gamma"], grads[ln + "/beta"] = dg, dbeta
                need_dx = xin.shape[-1] != 3 or conv != "conv2d"
                d, dw, db = ops.conv2d_same_bwd(xin, q(params[conv + "/kernel"]), dz, need_dx=need_dx)
                d = q(d) if need_dx else None
                grads[conv + "/kernel"], grads[conv + "/bias"] = dw, db
            elif kind == "ca":
                _, conv, xin, a, zpre = rec
                dz = ops.relu_bwd(d, a) if kink == 0.0 else d * (zpre > kink)
                d, dw, db = ops.conv2d_same_bwd(xin, q(params[conv + "/kernel"]), dz)
                d = q(d)
                grads[conv + "/kernel"], grads[conv + "/bias"] = dw, db
            elif kind == "caf":
                _, conv, xin, a, zpre = rec
                dz = ops.relu_bwd(d, a) if kink == 0.0 else d * (zpre > kink)
                dyb = q(ops.upconv_gather_bwd(dz, xin.shape[1], xin.shape[2]))
                d, dw = ops.upconv_bank_bwd(xin, q(params[conv + "/kernel"]), dyb)
                d = q(d)
                grads[conv + "/kernel"], grads[conv + "/bias"] = dw, dz.reshape(-1, dz.shape[-1]).sum(axis=0)
            elif kind == "concat":
                _, c1, lvl = rec
                dskips[lvl] = d[..., c1:]
                d = d[..., :c1]
            elif kind == "up":
                d = q(ops.resize_aa_bwd(d, rec[1], rec[1]))
            elif kind == "down":
                d = q(ops.resize_aa_bwd(d, rec[1], rec[1]) + dskips[rec[2]])
        return grads

    # ------------------------------------------------------------------ #
    def loss_and_grads(self, params, lr_img, hr_img, loss: str = "charbonnier", storage: "Storage | None" = None):
        out = self.forward(params, lr_img, storage=storage)
        if loss == "charbonnier":
            val = ops.charbonnier_fwd(hr_img, out)
            dout = ops.charbonnier_bwd(hr_img, out)
        elif loss == "l1":
            val = ops.l1_fwd(hr_img, out)
            dout = ops.l1_bwd(hr_img, out)
        else:
            raise ValueError(f"Unknown loss '{loss}'. Expected one of: 'charbonnier', 'l1', 'combined'.")
        grads = self.backward(params, dout)
        self._dout = dout
        psnr = float(np.mean(ops.psnr_per_image(hr_img, out)))
        return float(val), grads, out, psnr

    def kink_slack(self, params, delta: float = 1e-5):
        """After loss_and_grads: per-tensor bound on what ReLU / clip decisions within `delta` of their kink can change,
        max|g(+delta) - g(-delta)| (zero when no pre-activation is that close)."""
        hi = self.backward(params, self._dout, kink=delta)
        lo = self.backward(params, self._dout, kink=-delta)
        return {k: float(np.abs(hi[k] - lo[k]).max()) for k in hi}

    def train_step(self, params, opt_state, lr_img, hr_img, lr=1e-4, loss: str = "charbonnier",
                   storage: "Storage | None" = None, scaler=None):
        """One Keras train step (forward, loss, backward, Keras-form Adam). Mutates params/state.
        scaler: an oracle.loss_scale.DynamicLossScale (the mixed_float16 policy): the loss gradient is multiplied by the
        scale before the backward pass, a non-finite step is skipped, the update uses the unscaled gradients."""
        if scaler is not None:
            out = self.forward(params, lr_img, storage=storage)
            val = ops.charbonnier_fwd(hr_img, out) if loss == "charbonnier" else ops.l1_fwd(hr_img, out)
            dout = ops.charbonnier_bwd(hr_img, out) if loss == "charbonnier" else ops.l1_bwd(hr_img, out)
            # the product forms (1 / count) * scale in float32 and multiplies the loss derivative by it
            gs = float(np.float32(np.float32(1.0 / dout.size) * np.float32(scaler.scale)))
            with np.errstate(all="ignore"):
                scaled = self.backward(params, dout * dout.size * gs)
            psnr = float(np.mean(ops.psnr_per_image(hr_img, out)))
            inv = float(np.float32(1.0) / np.float32(scaler.scale))
            if not scaler.update(scaled):
                return float(val), psnr
            grads = {k: v * inv for k, v in scaled.items()}
        else:
            val, grads, out, psnr = self.loss_and_grads(params, lr_img, hr_img, loss, storage=storage)
        opt_state["step"] = opt_state.get("step", 0) + 1
        for name in params:
            m = opt_state.setdefault("m/" + name, np.zeros_like(params[name]))
            v = opt_state.setdefault("v/" + name, np.zeros_like(params[name]))
            ops.adam_step(params[name], grads[name], m, v, opt_state["step"], lr=lr)
        return val, psnr
