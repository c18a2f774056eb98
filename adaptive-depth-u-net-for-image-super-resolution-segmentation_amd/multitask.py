"""BASELINE config 5 -- "adaptive per-sample depth 2-6, mixed SR + seg multitask, fp16" -- as a driver over the models the
reference does have.

The reference has no per-sample depth and no multitask code: depth is a build-time integer of ONE static Keras graph, its
Experiment 2 trains an independent model per (scale, depth) row of a table
(/root/reference/Super_resolution/sbatch_scripts/run_experiment_adaptive_depth.sh:47-55), and the segmentation model is
trained by a separate script (Segmenation/code/train_adaptive_unet.py:463-575).  The semantics below are therefore
BUILD-DEFINED, chosen so that every sub-step is exactly a step the reference graph defines:

* a BANK of `build_super_resolution_unet(scale, depth_override=depth)` models, one per (scale, depth), built on first
  use; no weight is shared between them (as in Experiment 2);
* a sample's depth comes from its scale: the Experiment-2 table where it has a row, else `custom_depth_from_scale`
  (shared/custom_layers.py:42-75), clamped to the configured range (2-6);
* a mixed stream of SR samples is BUCKETED by (scale, depth): each bucket becomes one batch of one model (a batch never
  mixes graphs);
* the segmentation model (`build_adaptive_depth_unet`, protocol A or B) is a second task with its own weights and
  optimizer; tasks alternate in the order the caller's stream names them;
* precision: the reference's `mixed_float16` policy (fp16 storage + Keras dynamic loss scaling per model: every model's
  compile() wraps its Adam in a LossScaleOptimizer); bf16 / fp32 are accepted too;
* every (task, scale, depth, batch shape) gets its own captured hipGraph (Model._fit_step: at most four shapes per
  model), replayed from then on;
* data parallel (`AdaptiveDepthBank.data_parallel()`): every model of the bank gets its OWN `parallel.DataParallel` (its own
  flat gradient buffer, buckets and rank-0 weight broadcast) when it is built.  Models are built on first use, so every rank
  must walk the SAME sequence of (task, scale) items -- the batches are sharded over the ranks, never the stream.

Tested as properties only (tests/test_multitask_gpu.py): a routed step is BITWISE the stand-alone model's step, models do
not disturb each other, the routing table is the reference's.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .custom_layers import custom_depth_from_scale

# run_experiment_adaptive_depth.sh:47-55 (scale -> encoder depth of Experiment 2)
EXPERIMENT2_DEPTH = {0.2: 1, 0.3: 2, 0.4: 3, 0.5: 3, 0.6: 4, 0.7: 5, 0.8: 5}


def route_depth(scale: float, min_depth: int = 2, max_depth: int = 6, input_size: int = 256) -> int:
    """Depth of the model that takes a sample of this scale: the Experiment-2 row if there is one, else the reference's
    heuristic for the patch size; clamped to [min_depth, max_depth]."""
    key = round(float(scale), 2)
    depth = EXPERIMENT2_DEPTH.get(key)
    if depth is None:
        depth = custom_depth_from_scale(float(scale), max_depth=max_depth, base_resolution=input_size)
    return max(min_depth, min(int(depth), max_depth))


def bucket_by_depth(samples: Iterable[Tuple[float, np.ndarray, np.ndarray]], batch_size: int, min_depth: int = 2,
                    max_depth: int = 6, input_size: int = 256):
    """Groups a mixed stream of (scale, lr [P,P,3], hr [P,P,3]) samples into batches of ONE (scale, depth) each, in order
    of first appearance; yields ((scale, depth), lr [B,P,P,3], hr [B,P,P,3]) whenever a bucket fills, the remainders at
    the end (a smaller last batch per bucket, as tf.data's drop_remainder=False)."""
    buckets: "OrderedDict[Tuple[float, int], List[Tuple[np.ndarray, np.ndarray]]]" = OrderedDict()
    for scale, lr, hr in samples:
        key = (round(float(scale), 2), route_depth(scale, min_depth, max_depth, input_size))
        items = buckets.setdefault(key, [])
        items.append((lr, hr))
        if len(items) == batch_size:
            yield key, np.stack([a for a, _ in items]), np.stack([b for _, b in items])
            items.clear()
    for key, items in buckets.items():
        if items:
            yield key, np.stack([a for a, _ in items]), np.stack([b for _, b in items])


class AdaptiveDepthBank:
    """One SR model per (scale, depth) + one segmentation model; `train_on_batch(task, ...)` routes a batch to its model."""

    def __init__(self, input_size: int = 256, min_depth: int = 2, max_depth: int = 6, dtype=None, device=None, seed: int = 1234,
                 learning_rate: float = 1e-4, loss: str = "charbonnier", seg_protocol: str = "A", seg_depth: int = 4,
                 seg_base_channels: int = 64):
        import torch
        self.input_size, self.min_depth, self.max_depth = int(input_size), int(min_depth), int(max_depth)
        self.dtype = dtype if dtype is not None else torch.float16          # config 5 names fp16
        self.device, self.seed = device, seed
        self.learning_rate, self.loss = learning_rate, loss
        self.seg_protocol, self.seg_depth, self.seg_base_channels = seg_protocol, seg_depth, seg_base_channels
        self.sr: "OrderedDict[Tuple[float, int], object]" = OrderedDict()
        self.seg = None
        self.steps: Dict[Tuple, int] = {}
        self._dp_kwargs: Optional[dict] = None         # set by data_parallel(): every model gets its own DataParallel
        self.dps: List[object] = []

    # ---- data parallelism: one exchange object per model (each model has its own flat gradient buffer)
    def data_parallel(self, bucket_bytes: int = 32 << 20, group=None, native: Optional[bool] = None):
        """Attach one `parallel.DataParallel` to every model of the bank -- those that exist now and, as they are built,
        those that do not yet (same construction order on every rank: see the module docstring).  Returns self."""
        self._dp_kwargs = {"bucket_bytes": bucket_bytes, "group": group, "native": native}
        for model in self.models():
            self._attach_dp(model)
        return self

    def _attach_dp(self, model):
        if self._dp_kwargs is not None and getattr(model, "_dp", None) is None:
            from .parallel import DataParallel
            if hasattr(model, "_require_device"):
                model._require_device()                # the flat buffers must exist before rank 0's weights are broadcast
            self.dps.append(DataParallel(model, **self._dp_kwargs))

    def close(self):
        """Destroy the native communicators of the models' exchange objects (before the process group goes away)."""
        for dp in self.dps:
            dp.close()

    # ---- models, built on first use
    def _build_sr(self, key: Tuple[float, int]):
        from .model import Adam, build_losses_and_metrics, build_super_resolution_unet
        model, _ = build_super_resolution_unet(key[0], depth_override=key[1], input_size=self.input_size, dtype=self.dtype,
                                               device=self.device, seed=self.seed)
        loss, metrics = build_losses_and_metrics(self.loss)
        model.compile(optimizer=Adam(learning_rate=self.learning_rate), loss=loss, metrics=metrics, jit_compile=False)
        return model

    def _build_seg(self, steps_per_epoch: int, epochs: int):
        from . import seg_model as S
        proto = S.PROTOCOLS[self.seg_protocol]
        seg = S.build_adaptive_depth_unet(self.input_size, self.seg_base_channels, self.seg_depth, dtype=self.dtype,
                                          device=self.device, seed=self.seed)
        seg.compile(optimizer=S.build_optimizer(proto, steps_per_epoch, epochs), loss=proto.loss_builder())
        return seg

    def sr_model(self, scale: float):
        key = (round(float(scale), 2), route_depth(scale, self.min_depth, self.max_depth, self.input_size))
        model = self.sr.get(key)
        if model is None:
            model = self.sr[key] = self._build_sr(key)
            self._attach_dp(model)
        return key, model

    def seg_model(self, steps_per_epoch: int = 100, epochs: int = 1):
        if self.seg is None:
            self.seg = self._build_seg(steps_per_epoch, epochs)
            self._attach_dp(self.seg)
        return self.seg

    # ---- steps
    def train_on_batch(self, task: str, *payload, graphed: bool = True):
        """task "sr": payload (scale, lr, hr) -> (key, loss, psnr); task "seg": payload (image, mask) -> ("seg", loss, dice, iou).
        graphed: replay the model's captured hipGraph for this batch shape (captured at first use)."""
        if task == "sr":
            scale, lr, hr = payload
            key, model = self.sr_model(scale)
            vals = model._fit_step(lr, hr) if graphed else model.train_on_batch(lr, hr)
            self.steps[("sr",) + key] = self.steps.get(("sr",) + key, 0) + 1
            return (key,) + tuple(vals)
        if task == "seg":
            img, mask = payload
            model = self.seg_model()
            vals = model._fit_step(img, mask) if graphed else model.train_on_batch(img, mask)
            self.steps[("seg",)] = self.steps.get(("seg",), 0) + 1
            return ("seg",) + tuple(vals)
        raise ValueError(f"unknown task '{task}' (expected 'sr' or 'seg')")

    def fit(self, stream: Iterable[Tuple], steps: Optional[int] = None, graphed: bool = True) -> Dict[Tuple, List[float]]:
        """Consumes (task, *payload) items in order (the caller decides how tasks alternate); returns the loss history per
        (task, scale, depth)."""
        history: Dict[Tuple, List[float]] = {}
        for i, item in enumerate(stream):
            if steps is not None and i >= steps:
                break
            out = self.train_on_batch(item[0], *item[1:], graphed=graphed)
            tag = ("seg",) if item[0] == "seg" else ("sr",) + out[0]
            history.setdefault(tag, []).append(float(out[1]))
        return history

    def models(self) -> Sequence:
        return list(self.sr.values()) + ([self.seg] if self.seg is not None else [])
