"""Segmentation U-Nets on MI355X (tier-2 model family).

Mirrors of
* ``build_adaptive_depth_unet(input_size, base_channels, depth)`` -- /root/reference/Segmenation/code/
  train_adaptive_unet.py:335-362: [Conv3x3+bias -> BatchNorm -> ReLU]x2 blocks, MaxPooling2D(2),
  UpSampling2D(2, bilinear), Concatenate([up, skip]), Conv2D(1, 1, sigmoid) head "lesion_mask";
* ``build_unet(input_size, num_classes, base_channels, depth)`` -- Segmenation/code/unet_vinillia.py:72-91:
  LayerNorm blocks, Conv2DTranspose(nf, 2, strides=2) decoder, head "mask_logits";
* losses / metrics / protocols -- train_adaptive_unet.py:258-318, 382-403, 451-460.
All tensor work runs in the HIP kernels of csrc/ (conv3x3 MFMA kernels, tier2.hip); BatchNorm keeps per-replica
batch statistics under data parallelism exactly as Keras does (no sync-BN in the reference).
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops, resize_tables
from .model import Adam, ConvSpec, LayerRow, Model, _uname

DEFAULT_IMAGE_SIZE = 256      # Segmenation/code/train_adaptive_unet.py:54-56
DEFAULT_BASE_CHANNELS = 64
DEFAULT_DEPTH = 4


class _SegLoss:
    def __init__(self, name: str, bce_weight: float, dice_weight: float):
        self.__name__ = self.name = name
        self.bce_weight, self.dice_weight = bce_weight, dice_weight


def make_hybrid_ce_dice_loss(alpha: float, beta: float) -> _SegLoss:
    """train_adaptive_unet.py:283-292: alpha * BCE + beta * (1 - dice)."""
    return _SegLoss("hybrid_ce_dice", alpha, beta)


def make_bce_dice_loss(bce_weight: float, dice_weight: float) -> _SegLoss:
    """train_adaptive_unet.py:295-304."""
    return _SegLoss("bce_dice", bce_weight, dice_weight)


def binary_crossentropy() -> _SegLoss:
    """Plain BCE, the vanilla baseline's loss (unet_vinillia.py)."""
    return _SegLoss("binary_crossentropy", 1.0, 0.0)


class CosineDecay:
    """tf.keras.optimizers.schedules.CosineDecay(initial_lr, decay_steps, alpha=0)."""

    def __init__(self, initial_learning_rate: float, decay_steps: int, alpha: float = 0.0):
        self.lr0, self.steps, self.alpha = initial_learning_rate, max(int(decay_steps), 1), alpha

    def __call__(self, step: int) -> float:
        frac = min(step, self.steps) / self.steps
        return self.lr0 * ((1 - self.alpha) * 0.5 * (1 + math.cos(math.pi * frac)) + self.alpha)


@dataclass
class ProtocolConfig:
    key: str
    description: str
    loss_builder: object
    initial_lr: float
    epochs: int
    batch_size: int
    cosine_schedule: bool
    early_stopping_patience: Optional[int]


PROTOCOLS: Dict[str, ProtocolConfig] = {                        # train_adaptive_unet.py:382-403
    "A": ProtocolConfig("A", "MSCA-UNet hybrid loss (0.4·CE + 0.6·Dice) with cosine annealing",
                        lambda: make_hybrid_ce_dice_loss(alpha=0.4, beta=0.6), 1e-3, 100, 8, True, 15),
    "B": ProtocolConfig("B", "D2HU-Net BCE+Dice loss (0.5·BCE + 1.0·Dice)",
                        lambda: make_bce_dice_loss(bce_weight=0.5, dice_weight=1.0), 3e-4, 200, 16, False, None),
}


def build_optimizer(protocol: ProtocolConfig, steps_per_epoch: int, epochs: int) -> Adam:
    """train_adaptive_unet.py:451-460."""
    if protocol.cosine_schedule:
        return Adam(learning_rate=CosineDecay(protocol.initial_lr, epochs * max(steps_per_epoch, 1), alpha=0.0))
    return Adam(learning_rate=protocol.initial_lr)


BASELINE_METRICS = ("accuracy", "precision", "recall", "dice_coefficient")      # Segmenation/code/unet_vinillia.py:266-271


class SegModel(Model):
    """U-Net for binary masks; norm in {"bn", "ln"}, up in {"bilinear", "convT"}."""

    def __init__(self, input_size: int, base_channels: int, depth: int, norm: str, up: str, name: str, head_name: str,
                 dtype: torch.dtype = torch.bfloat16, device=None, seed: int = 1234, num_classes: int = 1):
        if depth < 1 or input_size <= 0:
            raise ValueError("depth and input_size must be positive")
        if num_classes < 1:
            raise ValueError("num_classes must be positive")
        self.num_classes = num_classes
        if input_size % (2 ** depth):
            raise ValueError(f"input_size {input_size} must be divisible by 2**depth = {2 ** depth}")
        self.input_size, self.base, self.depth, self.norm, self.up = input_size, base_channels, depth, norm, up
        self.name, self.head_name = name, head_name
        self.dtype = dtype
        self.device = torch.device(device) if device is not None else None
        self.seed = seed
        self.layers: List[LayerRow] = []
        self.index = OrderedDict()
        self.state_index = OrderedDict()          # non-trainable BatchNorm moving statistics
        self.convs: Dict[str, ConvSpec] = {}
        self._nparams = self._nstate = 0
        self.blocks: List[List[Tuple[ConvSpec, str]]] = []
        self.ups: List[Optional[str]] = []
        self._build_graph()
        self.P = self.G = self.M = self.V = self.S = None
        self._packs, self._tpacks = {}, {}
        self._ws = None
        self.optimizer = None
        self.loss = None
        self.metrics_names = ["loss", "dice", "iou"]
        self.stop_training = False
        self.grad_sync = self.grad_ready = None
        self.audit = None                 # parity instrumentation, see Model.audit
        self._up_tabs = {}

    # ------------------------------------------------------------------ graph
    def _build_graph(self):
        cnt: Dict[str, int] = {}
        p = self.input_size
        self.layers.append(LayerRow("isic_image" if self.norm == "bn" else "images", "InputLayer", (p, p, 3), 0, []))
        norm_base = "batch_normalization" if self.norm == "bn" else "layer_normalization"
        norm_type = "BatchNormalization" if self.norm == "bn" else "LayerNormalization"

        def block(cin, nf, hw, first_needs_dgrad=True):
            out = []
            for i in range(2):
                cs = self._add_conv(cnt, cin if i == 0 else nf, nf, hw, [], need_dgrad=first_needs_dgrad if i == 0 else True)
                nn = _uname(cnt, norm_base)
                self._register(nn + "/gamma", (nf,))
                self._register(nn + "/beta", (nf,))
                nparams = 2 * nf
                if self.norm == "bn":
                    for sname in ("/moving_mean", "/moving_variance"):
                        self.state_index[nn + sname] = (self._nstate, (nf,))
                        self._nstate += nf
                    nparams = 4 * nf
                self.layers.append(LayerRow(nn, norm_type, (hw, hw, nf), nparams, [cs.name]))
                self.layers.append(LayerRow(_uname(cnt, "activation"), "Activation", (hw, hw, nf), 0, [nn]))
                cs.ln = nn
                out.append((cs, nn))
            self.blocks.append(out)

        nf, cin, hw = self.base, 3, p
        for lvl in range(self.depth):
            block(cin, nf, hw, first_needs_dgrad=lvl > 0)
            self.layers.append(LayerRow(_uname(cnt, "max_pooling2d"), "MaxPooling2D", (hw // 2, hw // 2, nf), 0, []))
            cin, nf, hw = nf, nf * 2, hw // 2
        block(cin, nf, hw)
        for _ in range(self.depth):
            nf //= 2
            hw *= 2
            if self.up == "convT":
                t = _uname(cnt, "conv2d_transpose")
                self._register(t + "/kernel", (2, 2, nf, 2 * nf))
                self._register(t + "/bias", (nf,))
                self.layers.append(LayerRow(t, "Conv2DTranspose", (hw, hw, nf), 4 * nf * 2 * nf + nf, []))
                self.ups.append(t)
                cat_c = 2 * nf
            else:
                self.layers.append(LayerRow(_uname(cnt, "up_sampling2d"), "UpSampling2D", (hw, hw, 2 * nf), 0, []))
                self.ups.append(None)
                cat_c = 3 * nf
            self.layers.append(LayerRow(_uname(cnt, "concatenate"), "Concatenate", (hw, hw, cat_c), 0, []))
            block(cat_c, nf, hw)
        k = self.num_classes
        self._register(self.head_name + "/kernel", (1, 1, nf, k))
        self._register(self.head_name + "/bias", (k,))
        self.layers.append(LayerRow(self.head_name, "Conv2D", (hw, hw, k), nf * k + k, []))
        self.head_channels = nf

    def count_params(self) -> int:
        return self._nparams + self._nstate           # Keras counts the BN moving statistics as (non-trainable) params

    # ------------------------------------------------------------------ parameters
    def _require_device(self):
        if self.P is not None:
            return
        from . import _lib
        _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("adunet_amd needs an MI355X (no GPU visible); there is no CPU fallback")
        if self.device is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.P = torch.zeros(self._nparams, dtype=torch.float32, device=self.device)
        self.G, self.M, self.V = torch.zeros_like(self.P), torch.zeros_like(self.P), torch.zeros_like(self.P)
        self.S = torch.zeros(max(self._nstate, 1), dtype=torch.float32, device=self.device)
        self._ws = ops.Workspace(self.device)
        self.set_weights(self.initial_weights(np.random.default_rng(self.seed)))

    def initial_weights(self, rng: np.random.Generator, head_uniform: float = 0.0) -> Dict[str, np.ndarray]:
        out = {}
        for name, (_, shape) in self.index.items():
            if name.endswith("/kernel"):
                if "transpose" in name:       # Keras layout [kh, kw, Cout, Cin]: fan_in = 4*Cin... glorot on (rf*Cout, rf*Cin)
                    limit = math.sqrt(6.0 / (4 * shape[2] + 4 * shape[3]))
                else:
                    rf = shape[0] * shape[1]
                    limit = math.sqrt(6.0 / (shape[2] * rf + shape[3] * rf))
                out[name] = rng.uniform(-limit, limit, size=shape).astype(np.float32)
            elif name.endswith("/gamma"):
                out[name] = np.ones(shape, np.float32)
            else:
                out[name] = np.zeros(shape, np.float32)
        for name, (_, shape) in self.state_index.items():
            out[name] = (np.zeros if name.endswith("moving_mean") else np.ones)(shape, np.float32)
        return out

    def set_weights(self, weights: Dict[str, np.ndarray]):
        self._require_device()
        Model.set_weights(self, weights)
        if self._nstate:
            host = np.empty(self._nstate, np.float32)
            for name, (off, shape) in self.state_index.items():
                if name not in weights:
                    raise ValueError(f"missing weights: {name}")
                host[off:off + int(np.prod(shape))] = np.asarray(weights[name], np.float32).reshape(-1)
            self.S[:self._nstate].copy_(torch.from_numpy(host))

    def get_weights(self) -> Dict[str, np.ndarray]:
        out = Model.get_weights(self)
        host = self.S.cpu().numpy()
        for n, (o, s) in self.state_index.items():
            out[n] = host[o:o + int(np.prod(s))].reshape(s).copy()
        return out

    def _state(self, name: str) -> torch.Tensor:
        off, shape = self.state_index[name]
        return self.S[off:off + int(np.prod(shape))]

    def _repack(self):
        Model._repack(self)
        for t in self.ups:
            if t is not None:
                self._tpacks[t] = ops.conv_transpose2x2s2_pack(self.param(t + "/kernel"), self.dtype, out=self._tpacks.get(t))

    # ------------------------------------------------------------------ forward / backward
    def _up_tables(self, h: int, transposed: bool):
        key = (h, transposed)
        if key not in self._up_tabs:
            fn = resize_tables.aa_spans_transposed if transposed else resize_tables.aa_spans
            s, w = fn(h, 2 * h)
            self._up_tabs[key] = ops.ResampleTables(s, w, s, w, self.device)
        return self._up_tabs[key]

    def _block_fwd(self, blk, x1, x2, training, tape, keep, pool=False):
        """conv_block; pool=True: also MaxPooling2D(2) of the block's output -- returns (a, pooled), the pooling written by the
        last BatchNorm's apply pass where that kernel exists (training), else by ad_maxpool2_fwd."""
        pooled = None
        for li, (cs, nn) in enumerate(blk):
            g, b = self.param(nn + "/gamma"), self.param(nn + "/beta")
            # the network's first Conv2D on the raw fp32 image (3 -> 64): the dedicated 3-channel kernels, no zero-padded copy
            raw = x1.dtype == torch.float32 and x1.shape[-1] == 3 and self.dtype != torch.float32
            if self.norm == "bn":
                if raw:
                    z = ops.conv3x3_c3_fwd(x1, self.param(cs.name + "/kernel"), self.param(cs.name + "/bias"), dtype=self.dtype)
                else:
                    z = ops.conv3x3_fwd(x1, x2, self._packs[cs.name][0], self.param(cs.name + "/bias"), cs.cout)
                if training and pool and li == len(blk) - 1 and ops.batchnorm_pool_supported(z):
                    a, pooled, mean, rstd = ops.batchnorm_relu_pool_fwd_train(z, g, b, self._state(nn + "/moving_mean"),
                                                                              self._state(nn + "/moving_variance"), self._ws)
                elif training:
                    a, mean, rstd = ops.batchnorm_relu_fwd_train(z, g, b, self._state(nn + "/moving_mean"),
                                                                 self._state(nn + "/moving_variance"), self._ws)
                else:
                    a = ops.batchnorm_relu_fwd_infer(z, g, b, self._state(nn + "/moving_mean"), self._state(nn + "/moving_variance"))
                    mean = rstd = None
            elif raw:
                z, a, mean, rstd = ops.conv3x3_c3_ln_relu_fwd(x1, self.param(cs.name + "/kernel"), self.param(cs.name + "/bias"), g, b,
                                                              dtype=self.dtype, want_z=keep or self.audit is not None)
            else:
                z, a, mean, rstd = ops.conv3x3_ln_relu_fwd(x1, x2, self._packs[cs.name][0], self.param(cs.name + "/bias"),
                                                           g, b, cs.cout, want_z=keep or self.audit is not None)
            if keep:
                tape.append(("cna", cs, nn, x1, x2, z, mean, rstd))
            if self.audit is not None:
                self.audit.append(("fwd_cna", cs.name, x1, x2, z, a, mean, rstd, training))
            x1, x2 = a, None
        if pool:
            return x1, (pooled if pooled is not None else ops.maxpool2_fwd(x1))
        return x1

    def _forward_seg(self, img: torch.Tensor, mask: Optional[torch.Tensor], training: bool, keep: bool):
        tape, skips = [], []
        first = self.blocks[0][0][0]
        c3 = ops.conv3x3_c3_supported(img, first.cout, self.dtype) and os.environ.get("ADUNET_SEG_NO_C3") != "1"
        x = img if c3 else ops.pad_channels(img, ops.cin_granule(self.dtype), self.dtype)
        for lvl in range(self.depth):
            x, pooled = self._block_fwd(self.blocks[lvl], x, None, training, tape, keep, pool=True)
            skips.append(x)
            if keep:
                tape.append(("pool", x, lvl))
            if self.audit is not None:
                self.audit.append(("fwd_pool", f"pool{lvl}", x, pooled))
            x = pooled
        x = self._block_fwd(self.blocks[self.depth], x, None, training, tape, keep)
        for i, lvl in enumerate(reversed(range(self.depth))):
            t = self.ups[i]
            if t is not None:
                if keep:
                    tape.append(("convT", t, x))
                y = ops.conv_transpose2x2s2_fwd(x, self._tpacks[t][0], self.param(t + "/bias"), skips[lvl].shape[-1])
                if self.audit is not None:
                    self.audit.append(("fwd_convT", t, x, y))
                x = y
            else:
                if keep:
                    tape.append(("up2", x.shape[1]))
                y = ops.resample(x, self._up_tables(x.shape[1], False))
                if self.audit is not None:
                    self.audit.append(("fwd_up2", f"up{lvl}", x, y))
                x = y
            if keep:
                tape.append(("concat", lvl))
            x = self._block_fwd(self.blocks[self.depth + 1 + i], x, skips[lvl], training, tape, keep)
        if self.num_classes > 1:           # softmax head (unet_vinillia.py:89): probabilities only
            if keep or mask is not None:
                self._no_softmax_training()
            prob = ops.softmax_head_fwd(x, self.param(self.head_name + "/kernel").view(self.head_channels, self.num_classes),
                                        self.param(self.head_name + "/bias"))
            if self.audit is not None:
                self.audit.append(("fwd_softmax_head", self.head_name, x, prob))
            return prob, None, tape
        w = self.param(self.head_name + "/kernel").view(self.head_channels)
        prob, sums = ops.seg_head_fwd(x, w, self.param(self.head_name + "/bias"), mask, self._ws,
                                      counts=getattr(self, "_baseline_metrics", False))
        if isinstance(sums, tuple):              # (sums [n,3], counts [n,6]): the counts feed the vanilla baseline's metrics only
            sums, self._last_counts = sums
        if keep:
            tape.append(("head", x, prob, sums))
        if self.audit is not None:
            self.audit.append(("fwd_head", self.head_name, x, mask, prob, sums))
        return prob, sums, tape

    def _backward_seg(self, tape, mask):
        ws = self._ws
        audit = self.audit
        sc = self._scaler() if self.optimizer is not None else None
        dskips: Dict[int, torch.Tensor] = {}
        pending_skip = None
        d = None
        while tape:
            rec = tape.pop()
            kind = rec[0]
            if kind == "head":
                _, xh, prob, sums = rec
                d = ops.seg_head_bwd(xh, self.param(self.head_name + "/kernel").view(self.head_channels), mask, prob, sums,
                                     self.grad(self.head_name + "/kernel").view(self.head_channels),
                                     self.grad(self.head_name + "/bias"), self.loss.bce_weight, self.loss.dice_weight, ws,
                                     loss_scale=sc.state if sc is not None else None)
                self._done(self.head_name + "/kernel")
                if audit is not None:
                    audit.append(("bwd_head", self.head_name, xh, mask, prob, d))
            elif kind == "cna":
                _, cs, nn, x1, x2, z, mean, rstd = rec
                d_in = d
                g, b = self.param(nn + "/gamma"), self.param(nn + "/beta")
                if self.norm == "bn":      # the conv's bias gradient (column sums of dz as stored) comes out of the pass that writes dz
                    dz = ops.batchnorm_relu_bwd(d, z, mean, rstd, g, b, self.grad(nn + "/gamma"), self.grad(nn + "/beta"), ws,
                                                dbias=self.grad(cs.name + "/bias"))
                else:
                    dz = ops.layernorm_relu_bwd(d, z, mean, rstd, g, b, self.grad(nn + "/gamma"), self.grad(nn + "/beta"),
                                                self.grad(cs.name + "/bias"), ws)
                if x1.dtype == torch.float32 and x1.shape[-1] == 3 and self.dtype != torch.float32:     # the first layer on the raw image
                    ops.conv3x3_c3_wgrad(x1, dz, self.grad(cs.name + "/kernel"), ws)
                else:
                    ops.conv3x3_wgrad(x1, x2, dz, self.grad(cs.name + "/kernel"), cs.cin, ws)
                self._done(cs.name + "/kernel")
                dsk = None
                if not cs.need_dgrad:
                    d = None
                elif x2 is not None:
                    d, pending_skip = ops.conv3x3_fwd(dz, None, self._packs[cs.name][1], None, cs.cin, split=x1.shape[-1])
                    dsk = pending_skip
                else:
                    d = ops.conv3x3_fwd(dz, None, self._packs[cs.name][1], None, self._cin_pad(cs))
                if audit is not None:
                    audit.append(("bwd_cna", cs.name, x1, x2, z, mean, rstd, d_in, dz, d, dsk))
            elif kind == "concat":
                dskips[rec[1]] = pending_skip
            elif kind == "up2":
                d_in = d
                d = ops.resample(d, self._up_tables(rec[1], True))
                if audit is not None:
                    audit.append(("bwd_up2", "up", d_in, d))
            elif kind == "convT":
                _, t, xin = rec
                d_in = d
                d = ops.conv_transpose2x2s2_bwd(xin, d, self._tpacks[t][1], self.grad(t + "/kernel"), self.grad(t + "/bias"), ws)
                self._done(t + "/kernel")
                if audit is not None:
                    audit.append(("bwd_convT", t, xin, d_in, d))
            elif kind == "pool":
                _, xin, lvl = rec
                d_in = d
                skip_grad = dskips.pop(lvl)
                d = ops.maxpool2_bwd(d, xin, add=skip_grad)          # pooling gradient + skip gradient, one pass, one rounding
                if audit is not None:
                    audit.append(("bwd_pool", f"pool{lvl}", xin, d_in, skip_grad, d))

    # ------------------------------------------------------------------ Keras call surface
    def _to_dev_mask(self, a) -> torch.Tensor:
        t = torch.as_tensor(np.asarray(a, dtype=np.float32)) if not isinstance(a, torch.Tensor) else a
        if t.dim() != 4 or t.shape[-1] != 1:
            raise ValueError(f"expected a [B,H,W,1] mask batch, got {tuple(t.shape)}")
        return t.to(device=self.device, dtype=torch.float32).contiguous()

    def __call__(self, x, training: bool = False):
        self._require_device()
        as_numpy = not isinstance(x, torch.Tensor)
        prob, _, _ = self._forward_seg(self._to_dev(x), None, training=training, keep=False)
        return prob.cpu().numpy() if as_numpy else prob

    @staticmethod
    def _no_softmax_training():
        raise NotImplementedError(
            "num_classes > 1: the softmax head is built for inference (model(x) / predict); the reference defines no "
            "loss or metric for it -- every loss in Segmenation/code is binary (BCE / Dice on one channel) and every "
            "entry point passes num_classes=1 (unet_vinillia.py:263, unet_vinillia_optuna.py:142)")

    def compile(self, optimizer=None, loss=None, metrics=None, jit_compile: bool = False):
        if jit_compile:
            raise ValueError("jit_compile=True is not supported (the reference disables XLA as well)")
        if self.num_classes > 1:
            self._no_softmax_training()
        if loss is None or not hasattr(loss, "bce_weight"):
            raise ValueError("loss must come from make_hybrid_ce_dice_loss / make_bce_dice_loss / binary_crossentropy")
        self.optimizer = self._wrap_optimizer(optimizer if optimizer is not None else Adam())
        self.loss = loss
        self.metrics_names = ["loss", "dice", "iou"]
        # metrics=[...] naming accuracy / precision / recall / dice_coefficient (strings, or objects / functions with such a
        # `name` / `__name__`): the vanilla baseline's set, Segmenation/code/unet_vinillia.py:266-271
        names = [m if isinstance(m, str) else getattr(m, "name", getattr(m, "__name__", "")) for m in (metrics or [])]
        self._baseline_metrics = any(n in BASELINE_METRICS for n in names)
        if self._baseline_metrics:
            unknown = [n for n in names if n not in BASELINE_METRICS]
            if unknown:
                raise ValueError(f"unknown metrics {unknown}: the baseline set is {list(BASELINE_METRICS)}")
            self.metrics_names = ["loss"] + [n for n in BASELINE_METRICS if n in names]

    def _metrics_from(self, sums: torch.Tensor, count: float, smooth: float = 1e-6):
        if getattr(self, "_baseline_metrics", False):
            return self._baseline_metrics_from(sums, count, smooth)
        m = ops.seg_metrics(sums, count, self.loss.bce_weight, self.loss.dice_weight, smooth)        # :258-304
        return m[0], m[1], m[2]

    def _baseline_metrics_from(self, sums: torch.Tensor, count: float, smooth: float):
        """The vanilla baseline's per-batch values (unet_vinillia.py:266-271) followed by the running sums Keras' stateful
        metrics keep: (loss, accuracy, precision, recall, dice_coefficient | correct, elements, tp, predicted positives,
        positives).  `_reduce_logs` forms the epoch values: BinaryAccuracy = sum correct / sum elements, Precision = sum tp /
        sum predicted, Recall = sum tp / sum positives (all over the epoch, as Keras accumulates them), loss and the function
        metric dice_coefficient (global over the batch, smooth 1e-6, unclipped probability: :94-99) as batch means."""
        c = self._last_counts.sum(dim=0)                              # [tp, pp, pos, correct, sum y p, sum (y + p)]
        bce = sums[:, 0].sum() / count
        dice_b = ((2.0 * sums[:, 1] + smooth) / (sums[:, 2] + smooth)).mean()
        loss = self.loss.bce_weight * bce + self.loss.dice_weight * (1.0 - dice_b)
        tiny = 1e-7                                                   # Keras: divide_no_nan -> 0 when the denominator is 0
        acc = c[3] / count
        prec = torch.where(c[1] > 0, c[0] / torch.clamp(c[1], min=tiny), torch.zeros_like(c[0]))
        rec = torch.where(c[2] > 0, c[0] / torch.clamp(c[2], min=tiny), torch.zeros_like(c[0]))
        dice = (2.0 * c[4] + smooth) / (c[5] + smooth)
        vals = {"loss": loss, "accuracy": acc, "precision": prec, "recall": rec, "dice_coefficient": dice}
        cnt = torch.full_like(c[0], float(count))
        return tuple(vals[k] for k in self.metrics_names) + (c[3], cnt, c[0], c[1], c[2])

    def _reduce_logs(self, keys, totals, nbatches) -> dict:
        if not getattr(self, "_baseline_metrics", False):
            return Model._reduce_logs(self, keys, totals, nbatches)
        k = len(self.metrics_names)
        correct, elems, tp, pp, pos = (float(v) for v in totals[k:k + 5])
        out = {}
        for name, tot in zip(self.metrics_names, totals):
            if name == "accuracy":
                out[name] = correct / elems if elems > 0 else 0.0
            elif name == "precision":
                out[name] = tp / pp if pp > 0 else 0.0
            elif name == "recall":
                out[name] = tp / pos if pos > 0 else 0.0
            else:
                out[name] = float(tot) / nbatches
        return out

    def train_on_batch(self, img, mask):
        if self.optimizer is None:
            raise RuntimeError("call compile() first")
        self._require_device()
        x, m = self._to_dev(img), self._to_dev_mask(mask)
        prob, sums, tape = self._forward_seg(x, m, training=True, keep=True)
        self._backward_seg(tape, m)
        gscale = self.grad_sync(self) if self.grad_sync is not None else 1.0
        self._begin_step()
        self._apply_gradients(gscale)
        return self._metrics_from(sums, float(m.numel()))

    # ---- graph-replayed train step (Model.make_graphed_train_step / fit): same hooks, segmentation flavour
    def _graph_inputs(self, img, mask):
        return self._to_dev(img), self._to_dev_mask(mask)

    def _graph_forward_backward(self, sx, sm):
        prob, sums, tape = self._forward_seg(sx, sm, training=True, keep=True)
        self._backward_seg(tape, sm)
        return self._metrics_from(sums, float(sm.numel()))

    def _graph_extra_state(self):
        return [self.S]                     # BatchNorm moving statistics

    def test_on_batch(self, img, mask):
        self._require_device()
        x, m = self._to_dev(img), self._to_dev_mask(mask)
        _, sums, _ = self._forward_seg(x, m, training=False, keep=False)
        return self._metrics_from(sums, float(m.numel()))


def build_adaptive_depth_unet(input_size: int = DEFAULT_IMAGE_SIZE, base_channels: int = DEFAULT_BASE_CHANNELS,
                              depth: int = DEFAULT_DEPTH, *, dtype: torch.dtype = torch.bfloat16, device=None,
                              seed: int = 1234) -> SegModel:
    """Segmenation/code/train_adaptive_unet.py:335-362 (name adaptive_unet_depth{d}_c{c})."""
    return SegModel(input_size, base_channels, depth, "bn", "bilinear", f"adaptive_unet_depth{depth}_c{base_channels}",
                    "lesion_mask", dtype=dtype, device=device, seed=seed)


def build_unet(input_size: int, num_classes: int = 1, base_channels: int = 32, depth: int = 4, *,
               dtype: torch.dtype = torch.bfloat16, device=None, seed: int = 1234) -> SegModel:
    """Segmenation/code/unet_vinillia.py:72-91 (name unet_isic_baseline), reference default base_channels=32.
    num_classes > 1 gives the Conv2D(num_classes, 1, softmax) head of :89-90 for inference (`model(x)`, `predict`,
    weight loading); compile / train_on_batch / fit raise for it: the reference's losses and metrics are binary and
    none of its entry points passes anything but 1 (unet_vinillia.py:263, unet_vinillia_optuna.py:142)."""
    if num_classes < 1:
        raise ValueError("num_classes must be positive")
    gran = ops.cin_granule(dtype) if base_channels > 0 else 1
    if base_channels <= 0 or base_channels % gran:
        raise ValueError(f"base_channels must be a positive multiple of {gran} for {dtype} (MFMA contraction granule)")
    return SegModel(input_size, base_channels, depth, "ln", "convT", "unet_isic_baseline", "mask_logits",
                    dtype=dtype, device=device, seed=seed, num_classes=num_classes)
