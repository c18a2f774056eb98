"""Host-side mirror of /root/reference/shared/custom_layers.py.

Same names, arguments, error behaviour and ``get_config()`` keys as the reference's depth
heuristics (:10-82) and Keras layers ``ResizeByScale`` (:86-111), ``ResizeToMatch`` (:115-132),
``ClippedResidualAdd`` / ``ClipAdd`` (:135-142); the tensor work runs in the HIP kernel
``ad_resample`` (antialiased bilinear resize in fp32) instead of ``tf.image.resize``.
"""
from __future__ import annotations

from math import ceil
from typing import Dict, Optional, Tuple

import torch

from . import ops, resize_tables

SERIALIZED_NAMES = {
    "ResizeByScale": "resize>ResizeByScale",
    "ResizeToMatch": "resize>ResizeToMatch",
    "ClippedResidualAdd": "utils>ClippedResidualAdd",
}


def infer_depth_from_scale(scale: float, min_depth: int = 1, max_depth: int = 4) -> int:
    """custom_layers.py:10-28 -- scale <= 0.25 -> 1, <= 0.45 -> 2, else 3, clamped."""
    if not (0.05 < scale < 1.0):
        raise ValueError("Scale should be between 0 and 1 (exclusive).")
    if scale <= 0.25:
        depth = 1
    elif scale <= 0.45:
        depth = 2
    else:
        depth = 3
    return max(min_depth, min(depth, max_depth))


def depth_and_sizes(scale, min_res=21, max_depth=7):
    """custom_layers.py:31-40."""
    depth = 1
    sizes = [256]
    res = 256
    while res > min_res and depth < max_depth:
        res = ceil(res * scale)
        sizes.append(res)
        depth += 1
    return min(depth, max_depth), sizes


def custom_depth_from_scale(scale: float, min_depth: int = 1, max_depth: int = 7, *,
                            base_resolution: int = 256, min_feature: int = 21) -> int:
    """custom_layers.py:42-75 -- shrink until the next extent would drop below min_feature."""
    if not (0.05 < scale < 1.0):
        raise ValueError("Scale should be between 0 and 1 (exclusive).")
    if min_depth < 1:
        raise ValueError("min_depth must be at least 1.")
    if max_depth < 1:
        raise ValueError("max_depth must be at least 1.")
    if base_resolution <= 0:
        raise ValueError("base_resolution must be positive.")
    if min_feature < 1:
        raise ValueError("min_feature must be at least 1 pixel.")
    depth = max(min_depth, 1)
    extent = base_resolution
    while depth < max_depth:
        candidate = ceil(extent * scale)
        if candidate < min_feature:
            break
        extent = candidate
        depth += 1
    return max(min_depth, min(depth, max_depth))


def estimate_bottleneck_size(hr: int, scale: float, depth: int) -> int:
    """custom_layers.py:77-82 (diagnostic; uses round, unlike the real pyramid)."""
    size = hr
    for _ in range(depth):
        size = max(1, int(round(size * scale)))
    return size


class Layer:
    """Minimal Keras-Layer-shaped base: a name, ``__call__`` -> ``call``, ``get_config``."""

    _counters: Dict[str, int] = {}

    def __init__(self, name: Optional[str] = None, **kwargs):
        if kwargs:
            raise TypeError(f"unexpected keyword arguments {sorted(kwargs)}")
        self.name = name or type(self).__name__.lower()
        self.trainable = True
        self.dtype = "float32"

    def __call__(self, *args, **kwargs):
        return self.call(*args, **kwargs)

    def get_config(self) -> Dict[str, object]:
        return {"name": self.name, "trainable": self.trainable, "dtype": self.dtype}


class _ResizeBase(Layer):
    def __init__(self, method: str = "bilinear", antialias: bool = True, name: Optional[str] = None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.method = method
        self.antialias = antialias
        self._tables: Dict[Tuple, ops.ResampleTables] = {}

    def _check(self):
        if self.method != "bilinear" or not self.antialias:
            raise ValueError("only method='bilinear' with antialias=True (the reference's configuration) is implemented")

    def tables(self, h: int, w: int, oh: int, ow: int, device, transposed: bool = False) -> ops.ResampleTables:
        key = (h, w, oh, ow, str(device), transposed)
        tab = self._tables.get(key)
        if tab is None:
            fn = resize_tables.aa_spans_transposed if transposed else resize_tables.aa_spans
            sy, wy = fn(h, oh)
            sx, wx = fn(w, ow)
            tab = ops.ResampleTables(sy, wy, sx, wx, device)
            self._tables[key] = tab
        return tab

    def resize(self, x: torch.Tensor, oh: int, ow: int) -> torch.Tensor:
        self._check()
        return ops.resample(x, self.tables(x.shape[1], x.shape[2], oh, ow, x.device))

    def resize_grad(self, dy: torch.Tensor, h: int, w: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Gradient w.r.t. the [., h, w, .] input; added into `out` when given."""
        self._check()
        tab = self.tables(h, w, dy.shape[1], dy.shape[2], dy.device, transposed=True)
        return ops.resample(dy, tab, out=out, accumulate=out is not None)


class ResizeByScale(_ResizeBase):
    """custom_layers.py:86-111 -- antialiased bilinear resize to ceil(h*scale) x ceil(w*scale) (float32 ceil)."""

    def __init__(self, scale: float, method: str = "bilinear", antialias: bool = True, name: Optional[str] = None, **kwargs):
        super().__init__(method=method, antialias=antialias, name=name, **kwargs)
        self.scale = float(scale)

    def output_hw(self, h: int, w: int) -> Tuple[int, int]:
        return resize_tables.resized_extent(h, self.scale), resize_tables.resized_extent(w, self.scale)

    def call(self, x: torch.Tensor) -> torch.Tensor:
        nh, nw = self.output_hw(x.shape[1], x.shape[2])
        return self.resize(x, nh, nw)

    def get_config(self) -> Dict[str, object]:
        return {**super().get_config(), "scale": self.scale, "method": self.method, "antialias": self.antialias}


class ResizeToMatch(_ResizeBase):
    """custom_layers.py:115-132 -- resize x to the spatial size of ref."""

    def call(self, inputs: Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
        x, ref = inputs
        return self.resize(x, ref.shape[1], ref.shape[2])

    def get_config(self) -> Dict[str, object]:
        return {**super().get_config(), "method": self.method, "antialias": self.antialias}


class ClippedResidualAdd(Layer):
    """custom_layers.py:135-139 -- clip(float32(inp) + float32(residual), 0, 1).

    Inside the model this is fused with the 1x1 ``residual_rgb`` convolution (ad_head_fwd); called
    stand-alone it routes through the same kernel with an identity 1x1 kernel."""

    def call(self, inputs: Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
        inp, residual = inputs
        if inp.shape != residual.shape or inp.shape[-1] != 3:
            raise ValueError("ClippedResidualAdd expects two [N,H,W,3] tensors of equal shape")
        dev = inp.device
        ch = 16  # smallest head width the kernel supports in fp32: the residual is embedded in 16 channels
        cache = self.__dict__.setdefault("_consts", {})
        if dev not in cache:                                   # identity 1x1 kernel, zero bias, scratch: once per device
            eye = torch.zeros((ch, 3), dtype=torch.float32, device=dev)
            eye[0, 0] = eye[1, 1] = eye[2, 2] = 1.0
            cache[dev] = (eye, torch.zeros(3, dtype=torch.float32, device=dev), ops.Workspace(dev, 1 << 20))
        eye, zero_bias, ws = cache[dev]
        xh = ops.pad_channels(residual.to(torch.float32).contiguous(), ch, torch.float32)
        out, _, _ = ops.head_fwd(xh, eye, zero_bias, inp.to(torch.float32).contiguous(), None, ws)
        return out.to(inp.dtype)


ClipAdd = ClippedResidualAdd  # custom_layers.py:142 (legacy alias)
