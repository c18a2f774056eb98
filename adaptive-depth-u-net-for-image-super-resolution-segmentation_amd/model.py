"""Adaptive-depth SR U-Net on MI355X: model builder + Keras-shaped Model object.

Mirrors ``build_super_resolution_unet`` / ``conv_block`` / ``build_losses_and_metrics`` and the
``model.compile / fit / evaluate / __call__ / summary / load_weights`` surface used by
/root/reference/Super_resolution/code/train_adaptive_unet.py (:200-287, :294-373, :479-494,
:622-632, :677) and evaluate_model.py (:85-90, :107).  The graph is static (depth is a build-time
integer, exactly as in the reference); every tensor op runs in a hand-written HIP kernel through the
C ABI (include/adunet.h).  There is no CPU / PyTorch compute fallback.

HBM layout: activations NHWC in the compute dtype (bf16 or fp32); all trainable parameters live in
ONE flat fp32 buffer (plus flat gradient / Adam m / Adam v buffers of the same shape) so that the
optimizer is a single launch and data-parallel gradient exchange is a few large RCCL all-reduces;
conv kernels additionally keep MFMA-operand packs of their weights in the compute dtype.
"""
from __future__ import annotations

import os

import math
import time
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .custom_layers import (ClippedResidualAdd, ResizeByScale, ResizeToMatch, custom_depth_from_scale,
                            estimate_bottleneck_size)

DEFAULT_BASE_CHANNELS = 64            # train_adaptive_unet.py:56
DEFAULT_RESIDUAL_HEAD_CHANNELS = 64   # train_adaptive_unet.py:57


# --------------------------------------------------------------------------- #
# Losses / metrics / optimizer descriptors (train_adaptive_unet.py:294-373, :489-494)
# --------------------------------------------------------------------------- #
class _Loss:
    def __init__(self, name: str, kind: int, eps: float):
        self.__name__ = name
        self.name = name
        self.kind = kind
        self.eps = eps


class _Metric:
    def __init__(self, name: str):
        self.__name__ = name
        self.name = name


def build_losses_and_metrics(loss_name: str):
    """Same contract as train_adaptive_unet.py:294-373: returns (loss, [psnr_metric])."""
    key = loss_name.lower()
    if key == "charbonnier":
        return _Loss("charbonnier_loss", 0, 1e-3), [_Metric("psnr")]
    if key == "l1":
        return _Loss("l1_loss", 1, 0.0), [_Metric("psnr")]
    if key == "combined":
        raise NotImplementedError(
            "loss 'combined' needs VGG19(weights='imagenet') (train_adaptive_unet.py:337), a remote fetch; "
            "it is out of scope for the offline MI355X build")
    raise ValueError(f"Unknown loss '{loss_name}'. Expected one of: 'charbonnier', 'l1', 'combined'.")


class Adam:
    """tf.keras.optimizers.Adam defaults (epsilon 1e-7 outside the bias correction)."""

    def __init__(self, learning_rate: float = 1e-3, beta_1: float = 0.9, beta_2: float = 0.999, epsilon: float = 1e-7):
        self.learning_rate = learning_rate
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.iterations = 0

    def lr_at(self, step: int) -> float:
        lr = self.learning_rate
        return float(lr(step)) if callable(lr) else float(lr)


class LossScaleOptimizer:
    """tf.keras.mixed_precision.LossScaleOptimizer(inner, dynamic=True): what Keras wraps the optimizer in under the
    reference's mixed_float16 policy (Super_resolution/code/train_adaptive_unet.py:471-477).  Initial scale 2**15; a step
    whose gradients contain inf / NaN is skipped and halves the scale; `dynamic_growth_steps` (2000) consecutive finite
    steps double it; the inner optimizer's iteration count advances on applied steps only.

    The scaler state lives in device memory (include/adunet.h, ad_loss_scale_*): the backward pass, the finiteness
    check, the (possibly skipped) Adam update and the scale update run without a host round trip, so the step can be
    replayed from a hipGraph.  The host-side attributes are read back on demand (`sync()`)."""

    def __init__(self, inner_optimizer: Adam, initial_scale: float = 2.0 ** 15, dynamic_growth_steps: int = 2000):
        self.inner_optimizer = inner_optimizer
        self.initial_scale = float(initial_scale)
        self.dynamic_growth_steps = int(dynamic_growth_steps)
        self.state: Optional[torch.Tensor] = None      # device float32[8], see ad_loss_scale_check
        self.lr_dev: Optional[torch.Tensor] = None
        self.calls = 0                                 # train steps issued (applied + skipped)

    # the Adam hyper-parameters of the wrapped optimizer
    learning_rate = property(lambda self: self.inner_optimizer.learning_rate)
    beta_1 = property(lambda self: self.inner_optimizer.beta_1)
    beta_2 = property(lambda self: self.inner_optimizer.beta_2)
    epsilon = property(lambda self: self.inner_optimizer.epsilon)

    def lr_at(self, step: int) -> float:
        return self.inner_optimizer.lr_at(step)

    def ensure(self, device):
        if self.state is None:
            host = torch.zeros(8, dtype=torch.float32)
            host[0], host[1] = self.initial_scale, 1.0 / self.initial_scale
            host[4] = float(self.inner_optimizer.iterations)
            self.state = host.to(device)
            self.lr_dev = torch.zeros(1, dtype=torch.float32, device=device)

    def sync(self) -> Dict[str, float]:
        """Read the scaler back (one device -> host copy): dynamic scale, applied and skipped step counts."""
        st = self.state.cpu().tolist() if self.state is not None else [self.initial_scale, 0, 0, 0, self.inner_optimizer.iterations, 0]
        self.inner_optimizer.iterations = int(st[4])
        return {"loss_scale": st[0], "good_steps": int(st[2]), "applied": int(st[4]), "skipped": int(st[5])}

    @property
    def loss_scale(self) -> float:
        return self.sync()["loss_scale"]

    @property
    def iterations(self) -> int:
        return self.sync()["applied"]

    @iterations.setter
    def iterations(self, value: int):
        self.inner_optimizer.iterations = int(value)
        if self.state is not None:
            self.state[4] = float(value)


class History:
    def __init__(self):
        self.epoch: List[int] = []
        self.history: Dict[str, List[float]] = {}


# --------------------------------------------------------------------------- #
# Static graph description
# --------------------------------------------------------------------------- #
@dataclass
class ConvSpec:
    name: str
    cin: int
    cout: int
    hw: int
    k: int = 3
    ln: Optional[str] = None   # name of the LayerNormalization that follows (None: conv+ReLU or head)
    need_dgrad: bool = True


@dataclass
class LayerRow:
    name: str
    type: str
    shape: Tuple[int, int, int]
    params: int
    inbound: List[str] = field(default_factory=list)


def _uname(cnt: Dict[str, int], base: str) -> str:
    k = cnt.get(base, 0)
    cnt[base] = k + 1
    return base if k == 0 else f"{base}_{k}"


class Model:
    """Keras-Model-shaped object around the static SR U-Net graph."""

    def __init__(self, scale: float, depth: int, input_size: int, base_channels: int, head_channels: int,
                 dtype: torch.dtype = torch.bfloat16, device=None, seed: int = 1234):
        if depth < 1:
            raise ValueError("depth must be at least 1")
        if input_size <= 0:
            raise ValueError("input_size must be positive")
        self.scale, self.depth, self.input_size = float(scale), int(depth), int(input_size)
        self.base, self.head = int(base_channels), int(head_channels)
        self.name = f"U-Net_SR_scale{scale:.2f}_depth{depth}"          # train_adaptive_unet.py:279
        self.dtype = dtype
        self.device = torch.device(device) if device is not None else None
        self.seed = seed
        # ONE shared down layer and ONE shared up layer, as in the reference (:233-234)
        self.enc_down = ResizeByScale(scale, name="enc_down")
        self.dec_up = ResizeToMatch(name="dec_up")
        self.clip_add = ClippedResidualAdd(name="enhanced_rgb")
        self.layers: List[LayerRow] = []
        self.index: "OrderedDict[str, Tuple[int, Tuple[int, ...]]]" = OrderedDict()
        self.convs: Dict[str, ConvSpec] = {}
        self._plan: List[tuple] = []
        self._nparams = 0
        self._build_graph()
        # runtime state (allocated on first use)
        self.P = self.G = self.M = self.V = None
        self._packs: Dict[str, Tuple[torch.Tensor, Optional[torch.Tensor]]] = {}
        self._banks: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._ws: Optional[ops.Workspace] = None
        self.optimizer: Optional[Adam] = None
        self.loss = None
        self.metrics_names: List[str] = []
        self.stop_training = False
        self.grad_sync: Optional[Callable[["Model"], float]] = None   # data-parallel hook (parallel.py)
        self.grad_ready: Optional[Callable[[int], None]] = None       # called with the low offset of finished grads
        # Parity instrumentation: when set to a list, every forward / backward step appends (kind, name, tensors...) with
        # the device tensors it read and wrote, so a test can re-run each step's arithmetic on the product's OWN inputs
        # (tests/test_layerwise_gpu.py).  None (the default) costs nothing.
        self.audit: Optional[list] = None

    # ------------------------------------------------------------------ graph
    def _register(self, name: str, shape: Tuple[int, ...]):
        n = int(np.prod(shape))
        self.index[name] = (self._nparams, tuple(shape))
        self._nparams += n

    def _add_conv(self, cnt, cin, cout, hw, inbound, k=3, name=None, ln=False, need_dgrad=True) -> ConvSpec:
        name = name or _uname(cnt, "conv2d")
        self._register(name + "/kernel", (k, k, cin, cout))
        self._register(name + "/bias", (cout,))
        self.layers.append(LayerRow(name, "Conv2D", (hw, hw, cout), k * k * cin * cout + cout, inbound))
        cs = ConvSpec(name, cin, cout, hw, k=k, need_dgrad=need_dgrad)
        self.convs[name] = cs
        return cs

    def _add_block(self, cnt, cin, nf, hw, inbound, first_needs_dgrad=True) -> Tuple[List[ConvSpec], str]:
        specs = []
        prev = inbound
        for i in range(2):
            cs = self._add_conv(cnt, cin if i == 0 else nf, nf, hw, prev,
                                need_dgrad=first_needs_dgrad if i == 0 else True)
            ln = _uname(cnt, "layer_normalization")
            self._register(ln + "/gamma", (nf,))
            self._register(ln + "/beta", (nf,))
            self.layers.append(LayerRow(ln, "LayerNormalization", (hw, hw, nf), 2 * nf, [cs.name]))
            act = _uname(cnt, "activation")
            self.layers.append(LayerRow(act, "Activation", (hw, hw, nf), 0, [ln]))
            cs.ln = ln
            specs.append(cs)
            prev = [act]
        return specs, prev[0]

    def _build_graph(self):
        cnt: Dict[str, int] = {}
        p = self.input_size
        self.layers.append(LayerRow("low_res_input", "InputLayer", (p, p, 3), 0, []))
        nf, hw, cin, prev = self.base, p, 3, "low_res_input"
        self.sizes = [p]
        skip_names: List[str] = []
        plan: List[tuple] = []
        down_row = up_row = None
        for lvl in range(self.depth):                                   # encoder (:245-250)
            blk, prev = self._add_block(cnt, cin, nf, hw, [prev], first_needs_dgrad=lvl > 0)
            plan.append(("block", blk, None))
            skip_names.append(prev)
            nhw = self.enc_down.output_hw(hw, hw)[0]
            plan.append(("down", lvl, hw, nhw))
            if down_row is None:
                down_row = LayerRow("enc_down", "ResizeByScale", (nhw, nhw, nf), 0, [prev])
                self.layers.append(down_row)
            else:
                down_row.shape = (nhw, nhw, nf)
                down_row.inbound.append(prev)
            prev, hw, cin = "enc_down", nhw, nf
            self.sizes.append(hw)
            nf *= 2
        blk, prev = self._add_block(cnt, cin, nf, hw, [prev])            # bottleneck (:253)
        plan.append(("block", blk, None))
        for lvl in reversed(range(self.depth)):                         # decoder (:256-262)
            nf //= 2
            shw = self.sizes[lvl]
            plan.append(("up", lvl, hw, shw))
            if up_row is None:
                up_row = LayerRow("dec_up", "ResizeToMatch", (shw, shw, 2 * nf), 0, [prev, skip_names[lvl]])
                self.layers.append(up_row)
            else:
                up_row.shape = (shw, shw, 2 * nf)
                up_row.inbound += [prev, skip_names[lvl]]
            up = self._add_conv(cnt, 2 * nf, nf, shw, ["dec_up"])
            plan.append(("upconv", up, lvl))
            cat = _uname(cnt, "concatenate")
            self.layers.append(LayerRow(cat, "Concatenate", (shw, shw, 2 * nf), 0, [up.name, skip_names[lvl]]))
            blk, prev = self._add_block(cnt, 2 * nf, nf, shw, [cat])
            plan.append(("block", blk, lvl))
            hw = shw
        blk, prev = self._add_block(cnt, nf, self.head, hw, [prev])      # residual head (:265)
        plan.append(("block", blk, None))
        self._add_conv(cnt, self.head, 3, hw, [prev], k=1, name="residual_rgb")
        plan.append(("head",))
        self.layers.append(LayerRow("enhanced_rgb", "ClippedResidualAdd", (hw, hw, 3), 0,
                                    ["low_res_input", "residual_rgb"]))
        self._plan = plan

    # ------------------------------------------------------------------ Keras-shaped accessors
    def count_params(self) -> int:
        return self._nparams

    def summary(self, print_fn: Callable[[str], None] = print, line_length: int = 98):
        """Text table with the same columns as Keras' functional-model summary."""
        print_fn(f'Model: "{self.name}"')
        print_fn("_" * line_length)
        print_fn(f"{'Layer (type)':<38}{'Output Shape':<24}{'Param #':>12}  Connected to")
        print_fn("=" * line_length)
        for row in self.layers:
            shape = "(None, " + ", ".join(str(v) for v in row.shape) + ")"
            print_fn(f"{(row.name + ' (' + row.type + ')'):<38}{shape:<24}{row.params:>12,}  {', '.join(row.inbound)}")
        print_fn("=" * line_length)
        mb = self._nparams * 4 / 2 ** 20
        print_fn(f" Total params: {self._nparams:,} ({mb:.2f} MB)")
        print_fn(f" Trainable params: {self._nparams:,} ({mb:.2f} MB)")
        print_fn(" Non-trainable params: 0 (0.00 B)")

    # ------------------------------------------------------------------ parameters
    def _require_device(self):
        if self.P is not None:
            return
        from . import _lib
        _lib.load()  # raises loudly when the HIP library is missing
        if not torch.cuda.is_available():
            raise RuntimeError("adunet_amd needs an MI355X (no GPU visible); there is no CPU fallback")
        if self.device is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.P = torch.zeros(self._nparams, dtype=torch.float32, device=self.device)
        self.G = torch.zeros_like(self.P)
        self.M = torch.zeros_like(self.P)
        self.V = torch.zeros_like(self.P)
        self._ws = ops.Workspace(self.device)
        self.set_weights(self.initial_weights(np.random.default_rng(self.seed)))

    def initial_weights(self, rng: np.random.Generator, head_uniform: float = 0.0) -> Dict[str, np.ndarray]:
        """Keras initialisers: glorot-uniform kernels, zero biases, gamma 1 / beta 0, zero residual_rgb (:267-274)."""
        out = {}
        for name, (_, shape) in self.index.items():
            if name.endswith("/kernel"):
                if name.startswith("residual_rgb"):
                    out[name] = (rng.uniform(-head_uniform, head_uniform, size=shape).astype(np.float32)
                                 if head_uniform > 0 else np.zeros(shape, np.float32))
                else:
                    rf = shape[0] * shape[1]
                    limit = math.sqrt(6.0 / (shape[2] * rf + shape[3] * rf))
                    out[name] = rng.uniform(-limit, limit, size=shape).astype(np.float32)
            elif name.endswith("/gamma"):
                out[name] = np.ones(shape, np.float32)
            elif name.startswith("residual_rgb") and head_uniform > 0:
                out[name] = rng.uniform(-head_uniform, head_uniform, size=shape).astype(np.float32)
            else:
                out[name] = np.zeros(shape, np.float32)
        return out

    def _view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        off, shape = self.index[name]
        return buf[off:off + int(np.prod(shape))].view(shape)

    def param(self, name: str) -> torch.Tensor:
        return self._view(self.P, name)

    def grad(self, name: str) -> torch.Tensor:
        return self._view(self.G, name)

    def set_weights(self, weights: Dict[str, np.ndarray]):
        self._require_device()
        missing = [k for k in self.index if k not in weights]
        if missing:
            raise ValueError(f"missing weights: {missing[:4]}{'...' if len(missing) > 4 else ''}")
        host = np.empty(self._nparams, np.float32)
        for name, (off, shape) in self.index.items():
            w = np.asarray(weights[name], dtype=np.float32)
            if tuple(w.shape) != shape:
                raise ValueError(f"{name}: expected shape {shape}, got {tuple(w.shape)}")
            host[off:off + w.size] = w.reshape(-1)
        self.P.copy_(torch.from_numpy(host))
        self._repack()

    def get_weights(self) -> Dict[str, np.ndarray]:
        self._require_device()
        host = self.P.cpu().numpy()
        return OrderedDict((n, host[o:o + int(np.prod(s))].reshape(s).copy()) for n, (o, s) in self.index.items())

    def get_grads(self) -> Dict[str, np.ndarray]:
        host = self.G.cpu().numpy()
        return OrderedDict((n, host[o:o + int(np.prod(s))].reshape(s).copy()) for n, (o, s) in self.index.items())

    # ---- checkpoints.  A Keras `.keras` archive holds the weights AND the optimizer variables (ModelCheckpoint at
    # train_adaptive_unet.py:613-618 saves whole models); `model.load_weights` (:511) reads the weights only, Keras'
    # BackupAndRestore (:615) brings back everything.  Same split here: one .safetensors file, weights under their Keras
    # names, training state under `optimizer/...`.
    def get_training_state(self) -> Dict[str, np.ndarray]:
        """Adam moments (per parameter, Keras variable order), iteration count and the loss scaler, as host arrays."""
        self._require_device()
        out: Dict[str, np.ndarray] = {}
        m, v = self.M.cpu().numpy(), self.V.cpu().numpy()
        for n, (o, shp) in self.index.items():
            k = int(np.prod(shp))
            out["optimizer/m/" + n] = m[o:o + k].reshape(shp).copy()
            out["optimizer/v/" + n] = v[o:o + k].reshape(shp).copy()
        opt = self.optimizer
        if opt is not None:
            out["optimizer/iterations"] = np.array([int(opt.iterations)], np.int64)
            if isinstance(opt, LossScaleOptimizer):
                opt.ensure(self.device)
                out["optimizer/loss_scale_state"] = opt.state.cpu().numpy().copy()
                out["optimizer/loss_scale_calls"] = np.array([opt.calls], np.int64)
        return out

    def set_training_state(self, state: Dict[str, np.ndarray]):
        self._require_device()
        m, v = np.zeros(self._nparams, np.float32), np.zeros(self._nparams, np.float32)
        for n, (o, shp) in self.index.items():
            k = int(np.prod(shp))
            m[o:o + k] = np.asarray(state["optimizer/m/" + n], np.float32).reshape(-1)
            v[o:o + k] = np.asarray(state["optimizer/v/" + n], np.float32).reshape(-1)
        self.M.copy_(torch.from_numpy(m))
        self.V.copy_(torch.from_numpy(v))
        opt = self.optimizer
        if opt is not None and "optimizer/iterations" in state:
            if isinstance(opt, LossScaleOptimizer) and "optimizer/loss_scale_state" in state:
                opt.ensure(self.device)
                opt.state.copy_(torch.from_numpy(np.asarray(state["optimizer/loss_scale_state"], np.float32)))
                opt.calls = int(state["optimizer/loss_scale_calls"][0])
                opt.inner_optimizer.iterations = int(state["optimizer/iterations"][0])
            else:
                opt.iterations = int(state["optimizer/iterations"][0])

    def save_weights(self, path: str, include_optimizer: bool = True):
        """`.safetensors`: weights under their Keras variable names (+ `optimizer/...` training state).  `.weights.h5` /
        `.keras`: Keras 3's own layouts, written without h5py (keras_archive.py: container pinned against libhdf5, Keras' store naming restated) -- weights and
        BatchNorm moving statistics only, as `model.load_weights` reads them."""
        path = str(path)
        if path.endswith(".keras"):
            from . import keras_archive
            return keras_archive.save_keras(self, path)
        if path.endswith(".h5"):
            from . import keras_archive
            return keras_archive.save_weights_h5(self, path)
        from safetensors.numpy import save_file
        tensors = {k: np.ascontiguousarray(v) for k, v in self.get_weights().items()}
        if include_optimizer and self.optimizer is not None:
            tensors.update({k: np.ascontiguousarray(v) for k, v in self.get_training_state().items()})
        save_file(tensors, str(path), metadata={"model": self.name, "format": "adunet_amd-flat-v2",
                                                "compute_dtype": str(self.dtype).replace("torch.", "")})

    def save(self, path: str):
        """`model.save(path)` (Segmenation/code/unet_vinillia.py:292; ModelCheckpoint(save_weights_only=False), :276)."""
        self.save_weights(path)

    def load_weights(self, path: str, restore_optimizer: bool = False):
        """Weights by Keras variable name (as `model.load_weights`).  restore_optimizer: also the Adam moments, iteration
        count and loss scaler when the file has them, so that training continues its trajectory bit for bit.
        `.keras` / `.h5`: Keras-3 archives and weight files (train_adaptive_unet.py:511-516, evaluate_model.py:79-91), read
        without h5py by keras_archive.py -- weights only, which is all Keras' `load_weights` restores."""
        path = str(path)
        if path.endswith(".safetensors"):
            from safetensors.numpy import load_file
            blob = load_file(path)
            self.set_weights({k: v for k, v in blob.items() if not k.startswith("optimizer/")})
            if restore_optimizer and "optimizer/iterations" in blob:
                self.set_training_state(blob)
        elif path.endswith(".npz"):
            with np.load(path, allow_pickle=False) as z:
                self.set_weights({k: z[k] for k in z.files if not k.startswith("optimizer/")})
                if restore_optimizer and "optimizer/iterations" in z.files:
                    self.set_training_state({k: z[k] for k in z.files})
        elif path.endswith(".keras") or path.endswith(".h5"):
            from . import keras_archive
            keras_archive.load_into(self, path)
        else:
            raise RuntimeError(f"unsupported checkpoint format: {path}")

    def _cin_pad(self, cs: ConvSpec) -> int:
        g = ops.cin_granule(self.dtype)
        return (cs.cin + g - 1) // g * g

    def _repack(self):
        """Refresh the MFMA operand packs (compute dtype) from the fp32 master weights: one launch for all layers."""
        batch = self.__dict__.get("_pack_batch")
        if batch is None or batch.owner is not self.P:
            layers = [(cs.name, self.param(cs.name + "/kernel"), self._cin_pad(cs), cs.need_dgrad)
                      for cs in self.convs.values() if cs.k == 3]
            batch = self._pack_batch = ops.PackBatch(layers, self.dtype, self.device)
            batch.owner = self.P                        # the table holds pointers into this buffer
            self._packs.update(batch.packs)
            self.__dict__["_banks"] = {}
        batch.run()
        # up-convs that run in the factored form (csrc/upconv.hip) contract against the 1x1 bank operands instead
        banks = self.__dict__.setdefault("_banks", {})
        for name in self._factored_upconvs():
            banks[name] = ops.pw_bank_pack(self.param(name + "/kernel"), self.dtype, out=banks.get(name))

    # ---- the decoder's `dec_up -> Conv2D(nf, 3, same, relu)` (:258-259) without the up-resized tensor
    FACTORED_MIN_SOURCE = 16        # smallest low-resolution extent that takes the factored form (below: 3x3 kernels)

    def _factored_upconvs(self) -> List[str]:
        """Names of the up-convs that run as a bank of nine 1x1 convolutions on the low-resolution map + an interpolating
        gather.  A static property of the graph (extents and channel counts), the same for every batch size."""
        cached = self.__dict__.get("_factored_names")
        if cached is not None:
            return cached
        names = []
        if os.environ.get("ADUNET_NO_FACTORED_UPCONV") != "1":
            for step in self.__dict__.get("_plan", []):     # (the segmentation models have no such step)
                if step[0] != "upconv":
                    continue
                cs, lvl = step[1], step[2]
                src = self.sizes[lvl + 1]
                if (src >= int(os.environ.get("ADUNET_FACTORED_MIN_SOURCE", self.FACTORED_MIN_SOURCE)) and cs.cin % 128 == 0 and cs.cout % 64 == 0
                        and self._upconv_tables(src, cs.hw).ok and self._upconv_tables(src, cs.hw).gather_fwd_ok(cs.cout, self.dtype)):
                    names.append(cs.name)
        self._factored_names = names
        return names

    def _upconv_tables(self, src: int, dst: int) -> "ops.UpconvTables":
        tabs = self.__dict__.setdefault("_up_tables", {})
        key = (src, dst)
        if key not in tabs:
            tabs[key] = ops.UpconvTables(src, src, dst, dst, self.device)
        return tabs[key]

    # ------------------------------------------------------------------ forward / backward
    def _to_dev(self, a) -> torch.Tensor:
        t = torch.as_tensor(np.asarray(a, dtype=np.float32)) if not isinstance(a, torch.Tensor) else a
        if t.dim() != 4 or t.shape[-1] != 3:
            raise ValueError(f"expected a [B,H,W,3] batch, got {tuple(t.shape)}")
        return t.to(device=self.device, dtype=torch.float32).contiguous()

    def _head_fuses_with_ln(self, tape: List[tuple], xh: torch.Tensor) -> bool:
        """The layer feeding the head is Conv2D -> LayerNorm -> ReLU (:265) recorded on the tape: its LayerNorm / ReLU
        backward runs inside the head's backward pass (ad_head_ln_bwd), which also reports the loss and the metric."""
        nxt = tape[-1] if tape else None
        if xh is None:        # the forward pass did not store the head's input: only the fused backward can re-derive it
            assert nxt is not None and nxt[0] == "cla"
            return True
        return (nxt is not None and nxt[0] == "cla" and nxt[4].shape == xh.shape
                and os.environ.get("ADUNET_NO_HEAD_LN_FUSION") != "1")

    def _head_input_stays_in_registers(self, need_out: bool, keep: bool, target) -> bool:
        """Train steps whose head runs as ad_head_ln_bwd alone (no forward launch over the head, _forward): that kernel can
        re-derive the head's input from the last layer's z, so the layer need not write its activation (0.5 GB at K2')."""
        return (not need_out and keep and target is not None and self.audit is None
                and os.environ.get("ADUNET_NO_HEAD_LN_FUSION") != "1" and os.environ.get("ADUNET_HEAD_FWD_IN_TRAIN") != "1"
                and os.environ.get("ADUNET_KEEP_HEAD_ACT") != "1")

    def _forward(self, x: torch.Tensor, target: Optional[torch.Tensor], keep: bool, need_out: bool = True):
        """need_out=False (train steps: only loss and metric leave the step, :622-632): where the head's backward kernel can
        report them, no forward launch runs over the head -- `out` is None and `stats` is filled by _backward."""
        loss_kind = self.loss.kind if self.loss is not None else 0
        eps = self.loss.eps if self.loss is not None else 1e-3
        tape: List[tuple] = []
        first = next(iter(self.convs.values()))
        # bf16: the first conv (3 -> 64) runs on the raw fp32 batch through the dedicated 3-channel kernels; otherwise the
        # input is zero-padded to the conv channel granule once
        c3 = first.cin == 3 and first.ln is not None and ops.conv3x3_c3_supported(x, first.cout, self.dtype)
        cur1, cur2 = (x if c3 else ops.pad_channels(x, ops.cin_granule(self.dtype), self.dtype)), None
        skips: List[torch.Tensor] = []
        no_head_act = self._head_input_stays_in_registers(need_out, keep, target)
        for si, step in enumerate(self._plan):
            kind = step[0]
            if kind == "block":
                for cs in step[1]:
                    feeds_head = (no_head_act and cs is step[1][-1] and si + 1 < len(self._plan) and self._plan[si + 1][0] == "head"
                                  and not (c3 and cs is first) and cs.cout == self.head
                                  and ops.conv3x3_ln_stats_is_fused(cur1, cur2, cs.cout))
                    if c3 and cs is first:
                        z, a, mean, rstd = ops.conv3x3_c3_ln_relu_fwd(cur1, self.param(cs.name + "/kernel"),
                                                                      self.param(cs.name + "/bias"),
                                                                      self.param(cs.ln + "/gamma"), self.param(cs.ln + "/beta"),
                                                                      dtype=self.dtype, want_z=keep or self.audit is not None)
                    else:
                        z, a, mean, rstd = ops.conv3x3_ln_relu_fwd(cur1, cur2, self._packs[cs.name][0],
                                                                   self.param(cs.name + "/bias"), self.param(cs.ln + "/gamma"),
                                                                   self.param(cs.ln + "/beta"), cs.cout, want_act=not feeds_head,
                                                                   want_z=keep or self.audit is not None)
                    if keep:
                        tape.append(("cla", cs, cur1, cur2, z, mean, rstd, step[2]))
                    if self.audit is not None:
                        self.audit.append(("fwd_cla", cs.name, cur1, cur2, z, a, mean, rstd))
                    cur1, cur2 = a, None
            elif kind == "down":
                skips.append(cur1)
                if keep:
                    tape.append(("down", step[1], cur1.shape[1], cur1.shape[2]))
                small = self.enc_down(cur1)
                if self.audit is not None:
                    self.audit.append(("fwd_resize", "enc_down", cur1, small))
                cur1 = small
            elif kind == "up":
                pass            # the resize belongs to the up-conv that follows (it may never materialise)
            elif kind == "upconv":
                cs, lvl = step[1], step[2]
                m = cur1.shape[0] * cur1.shape[1] * cur1.shape[2]
                if (cs.name in self._banks and ops.pw_supported(m, cs.cin, 9 * cs.cout, self.dtype)
                        and ops.pw_supported(m, 9 * cs.cout, cs.cin, self.dtype)):
                    # factored form: nine 1x1 convolutions on the low-resolution map, then the interpolating gather
                    tab = self._upconv_tables(cur1.shape[1], cs.hw)
                    ybank = ops.pw_gemm(cur1, self._banks[cs.name][0], 9 * cs.cout)
                    u = ops.upconv_gather_fwd(ybank, self.param(cs.name + "/bias"), tab, relu=True)
                    if keep:
                        tape.append(("caf", cs, cur1, u, tab))
                    if self.audit is not None:
                        self.audit.append(("fwd_bank", cs.name, cur1, ybank))
                        self.audit.append(("fwd_gather", cs.name, ybank, u))
                else:
                    if keep:
                        tape.append(("up", cur1.shape[1], cur1.shape[2]))
                    big = self.dec_up((cur1, skips[lvl]))
                    if self.audit is not None:
                        self.audit.append(("fwd_resize", "dec_up", cur1, big))
                    cur1 = big
                    u = ops.conv3x3_fwd(cur1, None, self._packs[cs.name][0], self.param(cs.name + "/bias"), cs.cout, relu=True)
                    if keep:
                        tape.append(("ca", cs, cur1, u))
                    if self.audit is not None:
                        self.audit.append(("fwd_ca", cs.name, cur1, u))
                # L.Concatenate()([x, skip]) (:261) is virtual: the skip joins as the second operand of the next conv
                cur1, cur2 = u, skips[lvl]
            elif kind == "head":
                w = self.param("residual_rgb/kernel").view(self.head, 3)
                b = self.param("residual_rgb/bias")
                if cur1 is None or (not need_out and keep and target is not None and self.audit is None
                                    and self._head_fuses_with_ln(tape, cur1) and os.environ.get("ADUNET_HEAD_FWD_IN_TRAIN") != "1"):
                    stats = torch.empty(3, dtype=torch.float32, device=x.device)
                    sqerr = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
                    tape.append(("head", cur1, stats, sqerr))
                    return None, stats, sqerr, tape
                out, stats, sqerr = ops.head_fwd(cur1, w, b, x, target, self._ws, loss_kind=loss_kind, eps=eps)
                if keep:
                    tape.append(("head", cur1, None, None))
                if self.audit is not None:
                    self.audit.append(("fwd_head", "residual_rgb", cur1, x, target, out, stats))
                return out, stats, sqerr, tape
        raise AssertionError("plan without head")

    def _backward(self, tape: List[tuple], x: torch.Tensor, target: torch.Tensor, grad_scale: float):
        ws = self._ws
        audit = self.audit
        sc = self._scaler() if self.optimizer is not None else None
        dskips: Dict[int, torch.Tensor] = {}
        d = None
        relu_done = False
        dz_ready = None           # dz of the next "cla" record when the head's backward already produced it
        # (r04, measured and withdrawn: the weight gradients of the 16 x 16 ... 1 x 1 levels on a second HIP stream, forked /
        # joined with events inside the captured graph, own scratch buffer -- 11.24 against 11.08 ms per K2' step, three
        # alternating rounds on one box.  r02 had measured the same for all weight gradients; even these launches, which leave
        # most of the chip idle, do not fill each other's gaps through the graph's cross-stream dependencies.)
        while tape:
            rec = tape.pop()
            kind = rec[0]
            if kind == "head":
                w = self.param("residual_rgb/kernel").view(self.head, 3)
                nxt = tape[-1] if tape else None
                if self._head_fuses_with_ln(tape, rec[1]):
                    # the layer feeding the head is Conv2D -> LayerNorm -> ReLU (:265): its LayerNorm/ReLU backward runs
                    # inside the head's backward pass, the gradient of the head activations never goes to memory
                    _, hcs, _, _, hz, hmean, hrstd, _ = nxt
                    dz_ready = ops.head_ln_bwd(rec[1], w, self.param("residual_rgb/bias"), x, target, hz, hmean, hrstd,
                                               self.param(hcs.ln + "/gamma"), self.param(hcs.ln + "/beta"),
                                               self.grad("residual_rgb/kernel").view(self.head, 3), self.grad("residual_rgb/bias"),
                                               self.grad(hcs.ln + "/gamma"), self.grad(hcs.ln + "/beta"),
                                               self.grad(hcs.name + "/bias"), grad_scale, ws, loss_kind=self.loss.kind,
                                               eps=self.loss.eps, loss_scale=sc.state if sc is not None else None,
                                               stats=rec[2], sqerr=rec[3])
                    d = None
                    if audit is not None:
                        audit.append(("bwd_head_ln", "residual_rgb", rec[1], x, target, grad_scale, hcs.name, hz, hmean, hrstd,
                                      dz_ready))
                else:
                    assert rec[2] is None, "the forward pass left the loss to a fused head backward that did not run"
                    d = ops.head_bwd(rec[1], w, self.param("residual_rgb/bias"), x, target,
                                     self.grad("residual_rgb/kernel").view(self.head, 3), self.grad("residual_rgb/bias"),
                                     grad_scale, ws, loss_kind=self.loss.kind, eps=self.loss.eps,
                                     loss_scale=sc.state if sc is not None else None)
                    if audit is not None:
                        audit.append(("bwd_head", "residual_rgb", rec[1], x, target, grad_scale, d))
                self._done("residual_rgb/kernel")
            elif kind == "cla":
                _, cs, x1, x2, z, mean, rstd, lvl = rec
                d_in = d
                if dz_ready is not None:
                    dz, dz_ready = dz_ready, None
                else:
                    dz = ops.layernorm_relu_bwd(d, z, mean, rstd, self.param(cs.ln + "/gamma"), self.param(cs.ln + "/beta"),
                                                self.grad(cs.ln + "/gamma"), self.grad(cs.ln + "/beta"),
                                                self.grad(cs.name + "/bias"), ws)
                if x1.dtype == torch.float32 and self.dtype != torch.float32:      # raw 3-channel batch: first layer
                    ops.conv3x3_c3_wgrad(x1, dz, self.grad(cs.name + "/kernel"), ws)
                else:
                    ops.conv3x3_wgrad(x1, x2, dz, self.grad(cs.name + "/kernel"), cs.cin, ws)
                self._done(cs.name + "/kernel")
                dsk = None
                fused_relu = False
                if not cs.need_dgrad:
                    d = None
                elif x2 is not None:
                    # x1 is the up-conv's ReLU output (:259-261): where the library has the fused epilogue, this dgrad
                    # also applies that ReLU's gradient and sums the up-conv's bias gradient (no relu_bwd pass later)
                    up = tape[-1][1] if tape and tape[-1][0] in ("ca", "caf") and tape[-1][3] is x1 else None
                    fused_relu = (up is not None and os.environ.get("ADUNET_NO_DGRAD_RELU") != "1"
                                  and ops.conv3x3_dgrad_relu_is_fused(dz, cs.cin, x1.shape[-1]))
                    if fused_relu:
                        d, dsk = ops.conv3x3_dgrad_relu(dz, self._packs[cs.name][1], x1, self.grad(up.name + "/bias"), cs.cin, ws)
                    else:
                        d, dsk = ops.conv3x3_fwd(dz, None, self._packs[cs.name][1], None, cs.cin, split=x1.shape[-1])
                    dskips[lvl] = dsk
                else:
                    nxt = tape[-1] if tape else None
                    if (nxt is not None and nxt[0] == "cla" and nxt[4].shape == x1.shape and nxt[4].shape[-1] == cs.cin
                            and os.environ.get("ADUNET_NO_DGRAD_LN_FUSION") != "1"
                            and ops.conv3x3_dgrad_ln_bwd_is_fused(dz, cs.cin)):
                        # x1 is the activation of the Conv2D -> LayerNorm -> ReLU layer right below (conv_block's second
                        # conv, :200-210): that layer's LayerNorm / ReLU backward runs in this dgrad's epilogue, the
                        # gradient of the activation never goes to memory
                        _, hcs, _, _, hz, hmean, hrstd, _ = nxt
                        dz_ready = ops.conv3x3_dgrad_ln_bwd(dz, self._packs[cs.name][1], hz, hmean, hrstd,
                                                            self.param(hcs.ln + "/gamma"), self.param(hcs.ln + "/beta"),
                                                            self.grad(hcs.ln + "/gamma"), self.grad(hcs.ln + "/beta"),
                                                            self.grad(hcs.name + "/bias"), ws)
                        d = None
                        if audit is not None:
                            audit.append(("bwd_dgrad_ln", cs.name, dz, hcs.name, hz, hmean, hrstd, dz_ready))
                    else:
                        d = ops.conv3x3_fwd(dz, None, self._packs[cs.name][1], None, self._cin_pad(cs))
                relu_done = fused_relu
                if audit is not None:
                    # (the skip half is accumulated into in place later on: record a copy)
                    audit.append(("bwd_cla", cs.name, x1, x2, z, mean, rstd, d_in, dz, d,
                                  dsk.clone() if dsk is not None else None, fused_relu))
            elif kind == "ca":
                _, cs, xin, u = rec
                d_in = d
                # (already the pre-activation gradient, bias gradient included, when the dgrad above fused the ReLU)
                dz = d if relu_done else ops.relu_bwd(d, u, self.grad(cs.name + "/bias"), ws)
                relu_done = False
                ops.conv3x3_wgrad(xin, None, dz, self.grad(cs.name + "/kernel"), cs.cin, ws)
                self._done(cs.name + "/kernel")
                d = ops.conv3x3_fwd(dz, None, self._packs[cs.name][1], None, cs.cin)
                if audit is not None:
                    audit.append(("bwd_ca", cs.name, xin, u, d_in, dz, d))
            elif kind == "caf":
                # factored up-conv: dY = gather^T(dz), dx = dY Bank^T, dBank = x^T dY (all at the low resolution)
                _, cs, xin, u, tab = rec
                d_in = d
                dz = d if relu_done else ops.relu_bwd(d, u, self.grad(cs.name + "/bias"), ws)
                relu_done = False
                dyb = ops.upconv_gather_bwd(dz, tab)
                ops.upconv_bank_wgrad(xin, dyb, self.grad(cs.name + "/kernel"), ws)
                self._done(cs.name + "/kernel")
                d = ops.pw_gemm(dyb, self._banks[cs.name][1], cs.cin)
                if audit is not None:
                    audit.append(("bwd_caf", cs.name, xin, u, d_in, dz, dyb, d))
            elif kind == "up":
                d_in = d
                d = self.dec_up.resize_grad(d, rec[1], rec[2])
                if audit is not None:
                    audit.append(("bwd_resize", "dec_up", d_in, None, d))
            elif kind == "down":
                _, lvl, h, w = rec
                d_in, acc = d, dskips.pop(lvl)
                nxt = tape[-1] if tape else None
                tab = self.enc_down.tables(h, w, d.shape[1], d.shape[2], d.device, transposed=True)
                if (nxt is not None and nxt[0] == "cla" and nxt[4].shape == acc.shape and ops.resample_ln_bwd_supported(acc, tab)
                        and os.environ.get("ADUNET_NO_SKIP_LN_FUSION") != "1"):
                    # the skip's gradient junction and the LayerNorm/ReLU backward of the block that produced the skip in
                    # one pass: dskip + enc_down^T(d) stays in registers, dz of that block comes out
                    _, hcs, _, _, hz, hmean, hrstd, _ = nxt
                    dz_ready = ops.resample_ln_bwd(d, tab, acc, hz, hmean, hrstd, self.param(hcs.ln + "/gamma"),
                                                   self.param(hcs.ln + "/beta"), self.grad(hcs.ln + "/gamma"),
                                                   self.grad(hcs.ln + "/beta"), self.grad(hcs.name + "/bias"), ws)
                    d = None
                    if audit is not None:
                        audit.append(("bwd_resize_ln", "enc_down", d_in, acc, hcs.name, hz, hmean, hrstd, dz_ready))
                else:
                    before = acc.clone() if audit is not None else None    # the skip gradient is accumulated into in place
                    d = self.enc_down.resize_grad(d, h, w, out=acc)
                    if audit is not None:
                        audit.append(("bwd_resize", "enc_down", d_in, before, d))

    def _done(self, name: str):
        if self.grad_ready is not None:
            self.grad_ready(self.index[name][0])

    # ------------------------------------------------------------------ Keras call surface
    def __call__(self, x, training: bool = False):
        """model(lr, training=False) -> enhanced RGB [B,P,P,3] float32 (same container type as the input)."""
        self._require_device()
        as_numpy = not isinstance(x, torch.Tensor)
        out, _, _, _ = self._forward(self._to_dev(x), None, keep=False)
        return out.cpu().numpy() if as_numpy else out

    predict_on_batch = __call__

    def compile(self, optimizer=None, loss=None, metrics: Optional[Sequence] = None, jit_compile: bool = False):
        """model.compile(optimizer=Adam(lr), loss=..., metrics=[psnr], jit_compile=False) (:489-494)."""
        if jit_compile:
            raise ValueError("jit_compile=True is not supported (the reference disables XLA as well)")
        if isinstance(loss, str):
            loss, default_metrics = build_losses_and_metrics(loss)
            metrics = metrics if metrics is not None else default_metrics
        if loss is None or not hasattr(loss, "kind"):
            raise ValueError("loss must come from build_losses_and_metrics ('charbonnier' or 'l1')")
        self.optimizer = self._wrap_optimizer(optimizer if optimizer is not None else Adam())
        self.loss = loss
        self.metrics_names = ["loss"] + [getattr(m, "__name__", str(m)) for m in (metrics or [])]

    def _wrap_optimizer(self, optimizer):
        """Under a float16 policy Keras' compile() wraps the optimizer in a dynamic LossScaleOptimizer; so does this."""
        if self.dtype == torch.float16 and not isinstance(optimizer, LossScaleOptimizer):
            return LossScaleOptimizer(optimizer)
        return optimizer

    def _scaler(self) -> Optional[LossScaleOptimizer]:
        opt = self.optimizer
        if isinstance(opt, LossScaleOptimizer):
            opt.ensure(self.device)
            return opt
        return None

    def _apply_gradients(self, gscale: float, alpha_dev: Optional[torch.Tensor] = None):
        """Optimizer application shared by the eager and the graph-captured step.  With a LossScaleOptimizer:
        finiteness check -> Adam (skipped on overflow, gradients unscaled) -> scale update, all on the device."""
        opt = self.optimizer
        sc = self._scaler()
        if sc is not None:
            ops.loss_scale_check(self.G, sc.state)
            ops.adam_step_scaled(self.P, self.G, self.M, self.V, sc.lr_dev, sc.state, b1=opt.beta_1, b2=opt.beta_2,
                                 eps=opt.epsilon, gscale=gscale)
            ops.loss_scale_update(sc.state, sc.dynamic_growth_steps)
        elif alpha_dev is not None:
            ops.adam_step_dev(self.P, self.G, self.M, self.V, alpha_dev, b1=opt.beta_1, b2=opt.beta_2, eps=opt.epsilon,
                              gscale=gscale)
        else:
            ops.adam_step(self.P, self.G, self.M, self.V, opt.iterations, lr=opt.lr_at(opt.iterations - 1),
                          b1=opt.beta_1, b2=opt.beta_2, eps=opt.epsilon, gscale=gscale)
        self._repack()

    def _publish(self, dev_scalar: torch.Tensor, value: float):
        """value -> a 1-float device tensor as a stream-ordered host-to-device copy from a ring of page-locked slots (a slot is
        reused 256 steps later, long after its copy has run).  Until r04 this was `fill_`: a torch elementwise kernel per step,
        the one framework compute op beside the library's launches."""
        ring = getattr(self, "_pub_ring", None)
        if ring is None:
            ring = self._pub_ring = [torch.empty(256, dtype=torch.float32).pin_memory(), 0]
        slot = ring[1] % 256
        ring[1] += 1
        ring[0][slot] = value
        dev_scalar.copy_(ring[0][slot:slot + 1], non_blocking=True)

    def _begin_step(self, alpha_dev: Optional[torch.Tensor] = None):
        """Host-side bookkeeping before a step: advance the step count and publish the step size / learning rate."""
        opt = self.optimizer
        sc = self._scaler()
        if sc is not None:
            sc.calls += 1
            if callable(opt.learning_rate):
                # Keras evaluates a schedule at the inner optimizer's `iterations` = APPLIED steps (a step skipped on
                # overflow does not advance it): read the applied-step count back (one small device -> host copy per step,
                # only for scheduled learning rates under the float16 policy)
                self._publish(sc.lr_dev, opt.lr_at(int(sc.state[4].item())))
            else:
                self._publish(sc.lr_dev, opt.lr_at(0))
        else:
            opt.iterations += 1
            if alpha_dev is not None:
                self._publish(alpha_dev, ops.adam_alpha(opt.lr_at(opt.iterations - 1), opt.beta_1, opt.beta_2, opt.iterations))

    def forward_loss(self, lr_img, hr_img, keep: bool = False, need_out: bool = True):
        """Forward + fused loss.  Returns (out, loss_mean [device scalar], psnr_mean [device scalar], tape).
        need_out=False: a train step -- `out` may be None and the two scalars are then valid after _backward (see _forward)."""
        self._require_device()
        if self.loss is None:
            raise RuntimeError("call compile() first")
        x, t = self._to_dev(lr_img), self._to_dev(hr_img)
        if x.shape != t.shape:
            raise ValueError(f"input/target shape mismatch: {tuple(x.shape)} vs {tuple(t.shape)}")
        out, stats, sqerr, tape = self._forward(x, t, keep=keep, need_out=need_out)
        # stats = (loss sum, mean tf.image.psnr (:308-311), loss mean) straight from the head kernel: views, no torch op
        return out, stats[2], stats[1], (tape, x, t)

    def train_on_batch(self, lr_img, hr_img):
        """One Keras train step: forward, loss, backward, (gradient all-reduce), Keras-form Adam."""
        if self.optimizer is None:
            raise RuntimeError("call compile() first")
        out, loss, psnr, (tape, x, t) = self.forward_loss(lr_img, hr_img, keep=True, need_out=False)
        count = float(x.numel())
        self._backward(tape, x, t, 1.0 / count)
        gscale = self.grad_sync(self) if self.grad_sync is not None else 1.0
        self._begin_step()
        self._apply_gradients(gscale)
        return loss, psnr

    def test_on_batch(self, lr_img, hr_img):
        _, loss, psnr, _ = self.forward_loss(lr_img, hr_img, keep=False)
        return loss, psnr

    # ---- hooks of the graph-replayed step (overridden by the segmentation model)
    def _graph_inputs(self, a, b):
        """Device tensors of one batch, in the layout the captured step reads."""
        return self._to_dev(a), self._to_dev(b)

    def _graph_forward_backward(self, sx: torch.Tensor, st: torch.Tensor):
        """Forward, loss and backward on the static inputs; returns the device scalars the step reports."""
        out, loss, psnr, (tape, x, t) = self.forward_loss(sx, st, keep=True, need_out=False)
        self._backward(tape, x, t, 1.0 / float(x.numel()))
        return loss, psnr

    def _graph_extra_state(self) -> List[torch.Tensor]:
        """Tensors besides P / M / V that a train step changes (put back after a capture-only warm-up)."""
        return []

    def make_graphed_train_step(self, lr_example, hr_example, capture_only: bool = False):
        """Capture one full train step (forward, loss, backward, Adam, repack: ~190 launches) into hipGraphs and
        return `step(lr, hr) -> (loss, psnr)` that copies the batch into the graphs' static inputs and replays them.
        The launch-bound host loop disappears; the step-dependent Adam factor is fed through device memory.

        Under DataParallel the step is cut into segments at the points where a gradient bucket becomes final: the
        segments are replayed back to back on the compute stream and each bucket's RCCL all-reduce is launched eagerly
        (not captured) on the communication stream right after the segment that completes it, so the exchange of the
        decoder / bottleneck gradients (86 % of the bytes) overlaps the encoder's backward pass as in the eager path.

        The capture needs one eager warm-up step.  By default the example batch is then trained on twice (warm-up and
        first replay); with capture_only the weights, Adam moments and iteration count are put back afterwards, so the
        example batch has not been trained on when the function returns (fit() uses this)."""
        if self.optimizer is None:
            raise RuntimeError("call compile() first")
        dp = getattr(self, "_dp", None)
        if self.grad_sync is not None and dp is None:
            raise RuntimeError("graph capture needs the DataParallel object that installed the gradient hooks")
        self._require_device()
        sx, st = (t.clone() for t in self._graph_inputs(lr_example, hr_example))
        alpha_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
        opt = self.optimizer
        gscale = 1.0 / dp.world if dp is not None else 1.0
        segs: List[tuple] = []                   # (graph, buckets to all-reduce after it, wait for the exchange)
        cap = {"g": None, "pool": None, "next": 0}

        def seg_begin():
            g = torch.cuda.CUDAGraph()
            if cap["pool"] is None:
                cap["pool"] = torch.cuda.graph_pool_handle()
            # one pool: later segments see (and keep alive) the earlier ones' tensors.  thread_local: the process group's
            # watchdog thread polls events while we capture; in the default "global" mode its hipEventQuery fails with
            # hipErrorStreamCaptureUnsupported and takes the process down
            g.capture_begin(pool=cap["pool"], capture_error_mode="thread_local")
            cap["g"] = g

        def seg_end(buckets, sync=False):
            cap["g"].capture_end()
            segs.append((cap["g"], buckets, sync))
            cap["g"] = None

        def ready_while_capturing(low_offset: int):
            done = []
            while cap["next"] < len(dp.buckets) and dp.buckets[cap["next"]][0] >= low_offset:
                done.append(dp.buckets[cap["next"]])
                cap["next"] += 1
            if done:
                seg_end(done)
                seg_begin()

        def body(capturing: bool):
            outputs = self._graph_forward_backward(sx, st)
            if dp is not None:
                if capturing:
                    rest = dp.buckets[cap["next"]:]
                    cap["next"] = len(dp.buckets)
                    if rest:
                        seg_end(rest, sync=True)
                        seg_begin()
                    else:       # the last bucket closed with the last backward kernel: the open segment is still empty
                        segs[-1] = (segs[-1][0], segs[-1][1], True)
                else:
                    self.grad_sync(self)
            self._apply_gradients(gscale, alpha_dev)
            return outputs

        def set_alpha():
            self._begin_step(alpha_dev)

        extra = self._graph_extra_state()        # e.g. BatchNorm moving statistics (subclasses)
        sc = self._scaler()
        if sc is not None:
            extra = list(extra) + [sc.state]         # the scaler is part of what a capture-only warm-up must put back
        saved_state = ((self.P.clone(), self.M.clone(), self.V.clone(), (opt.iterations if sc is None else sc.calls),
                        [t.clone() for t in extra]) if capture_only else None)
        side = torch.cuda.Stream(device=self.device)       # warm-up on a side stream, as torch's capture recipe asks
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            set_alpha()
            body(False)                                     # sizes every workspace / attribute before capture
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        set_alpha()
        # No cyclic garbage collection while a capture is open: a collector run that happens to fall into the capture
        # destroys whatever unreachable cycles earlier code left behind (old hipGraphs with their memory pools, RCCL
        # communicators, streams), their destructors call APIs that are illegal during capture, and the exception inside a
        # destructor aborts the process (seen once as "Fatal Python error: Aborted ... Garbage-collecting").
        import gc
        gc_was_enabled = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            if dp is None:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):      # (a process group may exist: see seg_begin)
                    outputs = body(True)
                segs.append((graph, [], False))
            else:
                saved = self.grad_ready
                self.grad_ready = ready_while_capturing
                try:
                    with torch.cuda.stream(side):
                        seg_begin()
                        outputs = body(True)
                        seg_end([])
                finally:
                    self.grad_ready = saved
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
        finally:
            if gc_was_enabled:
                gc.enable()

        def run():
            for g, buckets, sync in segs:
                g.replay()
                for lo, hi in buckets:
                    dp._launch(lo, hi)
                if sync:
                    dp.wait_all()

        if capture_only:
            self.P.copy_(saved_state[0]); self.M.copy_(saved_state[1]); self.V.copy_(saved_state[2])
            if sc is None:
                opt.iterations = saved_state[3]
            else:
                sc.calls = saved_state[3]
            for t, saved in zip(extra, saved_state[4]):
                t.copy_(saved)
            self._repack()
            torch.cuda.synchronize()
        else:
            # the capture itself did not execute the step: run it once so that iteration counts stay truthful
            run()

        def step(lr_img, hr_img):
            a, b = self._graph_inputs(lr_img, hr_img)
            sx.copy_(a, non_blocking=True)
            st.copy_(b, non_blocking=True)
            set_alpha()
            run()
            return outputs

        step.graph = segs[0][0]
        step.segments = segs
        return step

    def _metric_keys(self) -> List[str]:
        return self.metrics_names if len(self.metrics_names) >= 2 else ["loss", "psnr"]

    def evaluate(self, dataset: Iterable, steps: Optional[int] = None, return_dict: bool = False, verbose: int = 0):
        keys = self._metric_keys()
        tot = None
        nb = 0
        for batch in dataset:
            vals = self.test_on_batch(batch[0], batch[1])
            tot = list(vals) if tot is None else [a + b for a, b in zip(tot, vals)]
            nb += 1
            if steps is not None and nb >= steps:
                break
        if nb == 0:
            raise ValueError("evaluate() received an empty dataset")
        res = self._reduce_logs(keys, tot, nb)
        return res if return_dict else [res[k] for k in keys]

    def _reduce_logs(self, keys, totals, nbatches) -> dict:
        """Epoch value of every metric from the per-batch values summed over the epoch: Keras' Mean (the loss and every
        function metric).  Models with ratio metrics (Precision / Recall: running sums) override this."""
        return {k: float(v) / nbatches for k, v in zip(keys, totals)}

    def _fit_step(self, lr_img, hr_img):
        """One training step of fit(): replayed from a hipGraph captured per batch shape on the GPU (bitwise the eager
        step, minus ~2 ms of Python launch overhead); eager on other devices or with ADUNET_EAGER_FIT=1."""
        if os.environ.get("ADUNET_EAGER_FIT") == "1" or (type(self).train_on_batch is not Model.train_on_batch
                                                         and type(self)._graph_forward_backward is Model._graph_forward_backward):
            return self.train_on_batch(lr_img, hr_img)      # a subclass with its own step and no graph hooks stays eager
        self._require_device()
        if self.device.type != "cuda":
            return self.train_on_batch(lr_img, hr_img)
        steps = self.__dict__.setdefault("_graph_steps", {})
        key = (tuple(lr_img.shape), self.optimizer, self.loss)
        step = steps.get(key)
        if step is None:
            if len(steps) >= 4:                     # e.g. a ragged last batch every epoch: do not hoard graph memory
                return self.train_on_batch(lr_img, hr_img)
            step = steps[key] = self.make_graphed_train_step(lr_img, hr_img, capture_only=True)
        return tuple(v.clone() for v in step(lr_img, hr_img))   # the graph's outputs are overwritten by the next replay

    def fit(self, dataset: Iterable, epochs: int = 1, initial_epoch: int = 0, steps_per_epoch: Optional[int] = None,
            validation_data: Optional[Iterable] = None, validation_steps: Optional[int] = None,
            callbacks: Optional[Sequence] = None, verbose: int = 2) -> History:
        """model.fit(train_ds, epochs, initial_epoch, steps_per_epoch, validation_data, validation_steps,
        callbacks, verbose) (:622-632).  `dataset` is any iterable of (lr, hr) float32 NHWC batches; it is
        consumed continuously across epochs when steps_per_epoch is given (as tf.data's infinite stream is)."""
        if self.optimizer is None:
            raise RuntimeError("call compile() first")
        callbacks = list(callbacks or [])
        hist = History()
        self.stop_training = False
        for cb in callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
            if hasattr(cb, "on_train_begin"):
                cb.on_train_begin({})
        for cb in callbacks:                 # BackupAndRestore: an interrupted fit() resumes after its last finished epoch
            if hasattr(cb, "restore_training_state"):
                initial_epoch = max(initial_epoch, cb.restore_training_state(self))
        it = iter(dataset)
        val_it = iter(validation_data) if validation_data is not None and validation_steps else None
        for epoch in range(initial_epoch, epochs):
            if verbose:
                print(f"Epoch {epoch + 1}/{epochs}", flush=True)
            t0 = time.time()
            keys = self._metric_keys()
            tot = None
            nb = 0
            while steps_per_epoch is None or nb < steps_per_epoch:
                try:
                    batch = next(it)
                except StopIteration:
                    if steps_per_epoch is None:
                        break                      # one pass over a finite dataset = one epoch
                    it = iter(dataset)             # keep streaming across epochs
                    try:
                        batch = next(it)
                    except StopIteration:
                        raise ValueError("fit() received an empty dataset") from None
                vals = self._fit_step(batch[0], batch[1])
                tot = list(vals) if tot is None else [a + b for a, b in zip(tot, vals)]
                nb += 1
            if steps_per_epoch is None:
                it = iter(dataset)
            if nb == 0:
                raise ValueError("fit() received an empty dataset")
            logs = self._reduce_logs(keys, tot, nb)
            if validation_data is not None:
                if val_it is not None:
                    vl = [self.test_on_batch(*next(val_it)[:2]) for _ in range(validation_steps)]
                    vres = self._reduce_logs(keys, [sum(v[i] for v in vl) for i in range(len(vl[0]))], len(vl))
                    for k in keys:
                        logs["val_" + k] = vres[k]
                else:
                    res = self.evaluate(validation_data, return_dict=True)
                    for k in keys:
                        logs["val_" + k] = res[k]
            dt = time.time() - t0
            if verbose:
                ms = dt * 1000.0 / nb
                step_txt = f"{ms:.0f}ms/step" if ms >= 1 else f"{ms * 1000:.0f}us/step"
                print(f"{nb}/{nb} - {dt:.0f}s - {step_txt} - " + " - ".join(f"{k}: {v:.4f}" for k, v in logs.items()),
                      flush=True)
            for cb in callbacks:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(epoch, logs)
            # Keras' History is itself a callback and runs LAST: keys a callback adds to `logs` (ReduceLROnPlateau's
            # `learning_rate`) are part of the history
            hist.epoch.append(epoch)
            for k, v in logs.items():
                hist.history.setdefault(k, []).append(v)
            if self.stop_training:
                break
        for cb in callbacks:
            if hasattr(cb, "on_train_end"):
                cb.on_train_end({})
        return hist


def build_super_resolution_unet(scale: float, base_channels: int = DEFAULT_BASE_CHANNELS,
                                residual_head_channels: int = DEFAULT_RESIDUAL_HEAD_CHANNELS,
                                depth_override: Optional[int] = None, input_size: int = 256, max_depth: int = 7, *,
                                dtype: torch.dtype = torch.bfloat16, device=None, seed: int = 1234):
    """Same signature and return value as train_adaptive_unet.py:217-287: (model, info).

    Extra keyword-only arguments choose the compute dtype (bf16 throughput path / fp32 parity path),
    the device and the initialiser seed."""
    depth = (depth_override if depth_override is not None
             else custom_depth_from_scale(scale, max_depth=max_depth, base_resolution=input_size))
    model = Model(scale, depth, input_size, base_channels, residual_head_channels, dtype=dtype, device=device, seed=seed)
    info = {
        "scale": scale,
        "depth": depth,
        "bottleneck_size": estimate_bottleneck_size(input_size, scale, depth),
        "base_channels": base_channels,
        "max_depth": max_depth,
    }
    return model, info
