"""NumPy oracle of the two segmentation U-Nets (test infrastructure only).

* ``norm="bn", up="bilinear"``: build_adaptive_depth_unet, /root/reference/Segmenation/code/train_adaptive_unet.py:325-362
  ([Conv3x3+bias -> BatchNorm -> ReLU]x2, MaxPool2, UpSampling2D(2, bilinear), Concatenate([up, skip]), sigmoid head).
* ``norm="ln", up="convT"``: build_unet, /root/reference/Segmenation/code/unet_vinillia.py:42-91
  (LayerNorm blocks, Conv2DTranspose(nf, 2, strides=2) decoder).
Loss = bce_weight * BCE + dice_weight * (1 - dice) (protocols A / B, :283-304, :382-403).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import ops


class SegUNetOracle:
    def __init__(self, input_size: int, base_channels: int = 64, depth: int = 4, norm: str = "bn", up: str = "bilinear",
                 num_classes: int = 1):
        if input_size % (2 ** depth):
            raise ValueError("input_size must be divisible by 2**depth")
        self.num_classes = num_classes       # > 1: softmax head (unet_vinillia.py:89-90), forward only
        self.p, self.base, self.depth, self.norm, self.up = input_size, base_channels, depth, norm, up
        self.param_shapes: Dict[str, tuple] = {}
        self.state_shapes: Dict[str, tuple] = {}
        self.blocks = []      # list of [(conv, norm), (conv, norm)]
        self.ups = []         # convT names per decoder level (or None)
        cnt = {}

        def uname(base):
            k = cnt.get(base, 0)
            cnt[base] = k + 1
            return base if k == 0 else f"{base}_{k}"

        def block(cin, nf):
            names = []
            for i in range(2):
                c = uname("conv2d")
                self.param_shapes[c + "/kernel"] = (3, 3, cin if i == 0 else nf, nf)
                self.param_shapes[c + "/bias"] = (nf,)
                nn = uname("batch_normalization" if norm == "bn" else "layer_normalization")
                self.param_shapes[nn + "/gamma"] = (nf,)
                self.param_shapes[nn + "/beta"] = (nf,)
                if norm == "bn":
                    self.state_shapes[nn + "/moving_mean"] = (nf,)
                    self.state_shapes[nn + "/moving_variance"] = (nf,)
                names.append((c, nn))
            self.blocks.append(names)

        nf, cin = base_channels, 3
        for _ in range(depth):
            block(cin, nf)
            cin, nf = nf, nf * 2
        block(cin, nf)
        for _ in range(depth):
            nf //= 2
            if up == "convT":
                t = uname("conv2d_transpose")
                self.param_shapes[t + "/kernel"] = (2, 2, nf, 2 * nf)
                self.param_shapes[t + "/bias"] = (nf,)
                self.ups.append(t)
                block(2 * nf, nf)
            else:
                self.ups.append(None)
                block(3 * nf, nf)      # concat([upsampled 2nf, skip nf])
        self.head = "lesion_mask" if norm == "bn" else "mask_logits"
        self.param_shapes[self.head + "/kernel"] = (1, 1, nf, num_classes)
        self.param_shapes[self.head + "/bias"] = (num_classes,)

    def count_params(self):
        return sum(int(np.prod(s)) for s in self.param_shapes.values())

    def init_params(self, rng, dtype=np.float64):
        params = {}
        for name, shape in self.param_shapes.items():
            if name.endswith("/kernel"):
                params[name] = ops.glorot_uniform(rng, shape, dtype) if len(shape) == 4 and shape[0] != 2 else \
                    rng.uniform(-0.1, 0.1, size=shape).astype(dtype)
            elif name.endswith("/gamma"):
                params[name] = rng.uniform(0.8, 1.2, size=shape).astype(dtype)
            else:
                params[name] = rng.uniform(-0.1, 0.1, size=shape).astype(dtype)
        state = {k: (np.zeros(s, dtype) if k.endswith("mean") else np.ones(s, dtype)) for k, s in self.state_shapes.items()}
        return params, state

    # ------------------------------------------------------------------
    def _block_fwd(self, x, names, params, state, training, tape, st):
        q = st.q
        for c, nn in names:
            w = q(params[c + "/kernel"])
            z = ops.conv2d_same_fwd(x, w, params[c + "/bias"])
            zs = q(z)                                # the conv output as stored
            g, b = params[nn + "/gamma"], params[nn + "/beta"]
            if self.norm == "bn":                    # BatchNorm is always its own kernel: statistics of the stored z
                if training:
                    y, (xhat, rstd), mu, var = ops.batchnorm_train_fwd(zs, g, b)
                    state[nn + "/moving_mean"] = state[nn + "/moving_mean"] * ops.BN_MOMENTUM + mu * (1 - ops.BN_MOMENTUM)
                    state[nn + "/moving_variance"] = state[nn + "/moving_variance"] * ops.BN_MOMENTUM + var * (1 - ops.BN_MOMENTUM)
                    cache = (zs, mu, rstd)
                else:
                    y, cache = ops.batchnorm_infer_fwd(zs, g, b, state[nn + "/moving_mean"], state[nn + "/moving_variance"]), None
            else:
                zin = z if st.fused(c, x.shape[0], x.shape[1], x.shape[2], x.shape[3], w.shape[3]) else zs
                y, (xhat, rstd) = ops.layernorm_fwd(zin, g, b)
                cache = (zs, zin.mean(axis=-1, keepdims=True), rstd)
            a = q(ops.relu_fwd(y))
            tape.append(("cna", c, nn, x, cache, a))
            x = a
        return x

    def forward(self, params, state, img, training=False, storage=None):
        """storage: oracle.sr_unet.Storage (None = exact arithmetic), see SRUNetOracle.forward."""
        from .sr_unet import Storage
        st = storage or Storage()
        q = st.q
        tape, skips = [], []
        x = q(img)
        for lvl in range(self.depth):
            x = self._block_fwd(x, self.blocks[lvl], params, state, training, tape, st)
            skips.append(x)
            tape.append(("pool", x, lvl))
            x = ops.maxpool2_fwd(x)
        x = self._block_fwd(x, self.blocks[self.depth], params, state, training, tape, st)
        for i, lvl in enumerate(reversed(range(self.depth))):
            if self.up == "convT":
                t = self.ups[i]
                tape.append(("convT", t, x))
                x = q(ops.conv_transpose2x2s2_fwd(x, q(params[t + "/kernel"]), params[t + "/bias"]))
            else:
                tape.append(("up2", x.shape[1]))
                x = q(ops.upsample2_bilinear_fwd(x))
            tape.append(("concat", x.shape[-1], lvl))
            x = np.concatenate([x, skips[lvl]], axis=-1)
            x = self._block_fwd(x, self.blocks[self.depth + 1 + i], params, state, training, tape, st)
        logit = ops.conv2d_same_fwd(x, params[self.head + "/kernel"], params[self.head + "/bias"])
        if self.num_classes > 1:             # activation="softmax" over the class axis; no loss is defined for it
            e = np.exp(logit - logit.max(axis=-1, keepdims=True))
            self._tape, self._storage = None, st
            return e / e.sum(axis=-1, keepdims=True)
        p = ops.sigmoid(logit)
        tape.append(("head", x, p))
        self._tape = tape
        self._storage = st
        return p

    def loss_and_grads(self, params, state, img, mask, bce_weight, dice_weight, storage=None):
        p = self.forward(params, state, img, training=True, storage=storage)
        loss, dp = ops.seg_loss_fwd_bwd(mask, p, bce_weight, dice_weight)
        self._dp = dp
        grads = self.backward(params, dp)
        dice = ops.dice_coefficient(mask, p)
        iou = ops.iou_score(mask, p)
        return float(loss), grads, p, float(dice), float(iou)

    def kink_slack(self, params, delta: float = 1e-5):
        """After loss_and_grads: per-tensor max|g(+delta) - g(-delta)| with every ReLU decision shifted by +-delta (what
        float32 rounding at a kink can change; see SRUNetOracle.kink_slack)."""
        hi, lo = self.backward(params, self._dp, kink=delta), self.backward(params, self._dp, kink=-delta)
        return {k: float(np.abs(hi[k] - lo[k]).max()) for k in hi}

    def backward(self, params, dp, kink: float = 0.0):
        q = self._storage.q
        grads = {}
        dskips = {}
        d = None
        for rec in reversed(self._tape):
            kind = rec[0]
            if kind == "head":
                _, xh, pp = rec
                d, dw, db = ops.conv2d_same_bwd(xh, params[self.head + "/kernel"], dp * pp * (1 - pp))
                d = q(d)
                grads[self.head + "/kernel"], grads[self.head + "/bias"] = dw, db
            elif kind == "cna":
                _, c, nn, xin, (zs, mu, rstd), a = rec
                xhat = (zs - mu) * rstd               # from what was saved: stored z, mean, rstd (ReLU mask re-derived)
                dy = d * (xhat * params[nn + "/gamma"] + params[nn + "/beta"] > kink)
                if self.norm == "bn":
                    dz, dg, dbeta = ops.batchnorm_train_bwd(dy, params[nn + "/gamma"], (xhat, rstd))
                else:
                    dz, dg, dbeta = ops.layernorm_bwd(dy, params[nn + "/gamma"], (xhat, rstd))
                dz = q(dz)
                grads[nn + "/gamma"], grads[nn + "/beta"] = dg, dbeta
                need_dx = xin.shape[-1] != 3
                d, dw, db = ops.conv2d_same_bwd(xin, q(params[c + "/kernel"]), dz, need_dx=need_dx)
                d = q(d) if need_dx else None
                grads[c + "/kernel"], grads[c + "/bias"] = dw, db
            elif kind == "concat":
                _, c1, lvl = rec
                dskips[lvl] = d[..., c1:]
                d = d[..., :c1]
            elif kind == "up2":
                d = q(ops.resize_aa_bwd(d, rec[1], rec[1]))
            elif kind == "convT":
                _, t, xin = rec
                d, dw, db = ops.conv_transpose2x2s2_bwd(xin, q(params[t + "/kernel"]), d)
                d = q(d)
                grads[t + "/kernel"], grads[t + "/bias"] = dw, db
            elif kind == "pool":
                d = q(ops.maxpool2_bwd(d, rec[1]) + dskips[rec[2]])
        return grads
