"""Thin torch-tensor wrappers over the C ABI (include/adunet.h).

PyTorch is used only to own device memory and streams; every computation below is a HIP kernel of
csrc/.  All activations are NHWC, dtype torch.bfloat16 (throughput) or torch.float32 (parity).
"""
from __future__ import annotations

import os

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import AD_BF16, AD_F16, AD_F32, EPI_NONE, EPI_RELU, check

LN_EPS = 1e-3          # Keras LayerNormalization default (train_adaptive_unet.py:203)
CHARBONNIER_EPS = 1e-3  # train_adaptive_unet.py:314


def dt(t: torch.dtype) -> int:
    if t == torch.bfloat16:
        return AD_BF16
    if t == torch.float32:
        return AD_F32
    if t == torch.float16:
        return AD_F16
    raise ValueError(f"unsupported activation dtype {t}")


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device, contiguous tensors only"
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional per-op-family timing with HIP events on the launch stream (used by bench.py).

    Every wrapper below brackets its launches with a pair of events when a timer is installed; the events
    sit on torch's current stream, which is the stream the kernels are launched on."""

    def __init__(self):
        self.records = {}

    def add(self, name, e0, e1, work=0.0, nbytes=0.0):
        self.records.setdefault(name, []).append((e0, e1, work, nbytes))

    def summary(self):
        """name -> (launches, total_ms); call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(r[0].elapsed_time(r[1]) for r in v)) for k, v in self.records.items()}

    def work(self, name) -> float:
        """Sum of the work figures (FLOPs as launched, padded channels included) the wrappers attached to `name`."""
        return sum(r[2] for r in self.records.get(name, []))

    def nbytes(self, name) -> float:
        """Sum of the algorithmic bytes (each operand tensor once) the wrappers attached to `name`."""
        return sum(r[3] for r in self.records.get(name, []))


_timer: Optional[KernelTimer] = None


def set_timer(t: Optional[KernelTimer]):
    global _timer
    _timer = t


class _timed:
    def __init__(self, name, work=0.0, nbytes=0.0):
        self.name = name
        self.work = work
        self.nbytes = nbytes

    def __enter__(self):
        if _timer is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if _timer is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _timer.add(self.name, self.e0, e1, self.work, self.nbytes)
        return False


def cin_granule(dtype: torch.dtype) -> int:
    return _lib.load().ad_cin_granule(dt(dtype))


class Workspace:
    """One scratch buffer reused by every call (the ABI never allocates).  Every use is confined to one ABI call (a
    kernel writes partials, the next launch of the same call folds them), so nothing is carried between calls.

    A captured hipGraph bakes the buffer's address into its kernel arguments.  When a later, larger request makes the
    buffer grow, the previous allocation is therefore RETIRED, not freed: it stays owned by this object, so replays of
    graphs captured earlier keep writing scratch into memory nobody else can receive from the caching allocator.  Growth
    is geometric, so the retired buffers sum to less than four times the live one."""

    def __init__(self, device, nbytes: int = 64 << 20):
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self._retired = []

    def ensure(self, nbytes: int):
        if self.buf.numel() < nbytes:
            self._retired.append(self.buf)
            self.buf = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=self.buf.device)

    @property
    def ptr(self):
        return self.buf.data_ptr()

    @property
    def nbytes(self):
        return self.buf.numel()


def pad_channels(x: torch.Tensor, cpad: int, dtype: torch.dtype) -> torch.Tensor:
    n, h, w, c = x.shape
    assert x.dtype == torch.float32
    y = torch.empty((n, h, w, cpad), dtype=dtype, device=x.device)
    with _timed("pad_channels", 0.0, float(x.numel() * 4 + y.numel() * y.element_size())):
        check(_lib.load().ad_pad_channels(_p(x), _p(y), n * h * w, c, cpad, dt(dtype), _stream()), "ad_pad_channels")
    return y


def conv3x3_pack(w_hwio: torch.Tensor, cin_pad: int, dtype: torch.dtype, want_dgrad: bool = True, out=None):
    """fp32 Keras kernel [3,3,cin,cout] -> (w_fwd, w_dgrad) operand tensors.  `out` = a previous result to refresh
    in place (the packs keep their addresses, which a captured hipGraph relies on)."""
    kh, kw, cin, cout = w_hwio.shape
    assert (kh, kw) == (3, 3) and w_hwio.dtype == torch.float32
    lib = _lib.load()
    if out is not None:
        wf, wd = out
    else:      # the packs pad their output-channel dimension to whole 64-channel blocks
        wf = torch.empty(lib.ad_conv3x3_pack_elems(cin_pad, cout, 0), dtype=dtype, device=w_hwio.device)
        wd = torch.empty(lib.ad_conv3x3_pack_elems(cin_pad, cout, 1), dtype=dtype, device=w_hwio.device) if want_dgrad else None
    with _timed("conv3x3_pack"):
        check(_lib.load().ad_conv3x3_pack(_p(w_hwio), cin, cout, cin_pad, _p(wf), _p(wd), dt(dtype), _stream()),
              "ad_conv3x3_pack")
    return wf, wd


class PackBatch:
    """All conv layers of a model repacked by one launch (ad_conv3x3_pack_batch).  Allocates the operand tensors once;
    `packs[name] = (w_fwd, w_dgrad)` keep their addresses, the job table lives in device memory."""

    def __init__(self, layers, dtype: torch.dtype, device):
        """layers: iterable of (name, w_hwio fp32 view [3,3,cin,cout], cin_pad, want_dgrad)."""
        lib = _lib.load()
        rec = np.dtype([("w", "<u8"), ("wf", "<u8"), ("wd", "<u8"), ("cin", "<i4"), ("cout", "<i4"), ("cin_pad", "<i4"),
                        ("first_block", "<i4")])
        assert rec.itemsize == lib.ad_conv3x3_pack_job_bytes()
        nblocks = 0
        self.packs = {}
        self._keep = []
        rows = []
        for name, w, cin_pad, want_dgrad in layers:
            kh, kw, cin, cout = w.shape
            assert (kh, kw) == (3, 3) and w.dtype == torch.float32 and w.is_contiguous()
            nf, nd = lib.ad_conv3x3_pack_elems(cin_pad, cout, 0), (lib.ad_conv3x3_pack_elems(cin_pad, cout, 1) if want_dgrad else 0)
            wf = torch.empty(nf, dtype=dtype, device=device)
            wd = torch.empty(nd, dtype=dtype, device=device) if want_dgrad else None
            self.packs[name] = (wf, wd)
            self._keep.append(w)
            rows.append((w.data_ptr(), wf.data_ptr(), wd.data_ptr() if wd is not None else 0, cin, cout, cin_pad, nblocks))
            nblocks += lib.ad_conv3x3_pack_job_blocks(cin_pad, cout)
        self.njobs = len(rows)
        self.nblocks = nblocks
        self.dtype = dtype
        table = np.array(rows, dtype=rec)
        self.table = torch.from_numpy(table.view(np.uint8).copy()).to(device)

    def run(self):
        with _timed("conv3x3_pack"):
            check(_lib.load().ad_conv3x3_pack_batch(_p(self.table), self.njobs, self.nblocks, dt(self.dtype), _stream()),
                  "ad_conv3x3_pack_batch")


_conv_ws: dict = {}   # per-device scratch for the split-K path of tiny feature maps


def _conv_workspace(device, nbytes: int):
    ws = _conv_ws.get(device)
    if ws is None:
        ws = _conv_ws[device] = Workspace(device, max(nbytes, 32 << 20))
    ws.ensure(nbytes)
    return ws


def conv3x3_fwd(x1: torch.Tensor, x2: Optional[torch.Tensor], w_packed: torch.Tensor, bias: Optional[torch.Tensor],
                cout: int, relu: bool = False, split: Optional[int] = None):
    """y = conv3x3_same(concat(x1, x2)) (+bias) (+ReLU).  With `split`, returns (y[..., :split], y[..., split:])
    as two separate tensors (dgrad of a concatenated input)."""
    n, h, w, c1 = x1.shape
    c2 = x2.shape[-1] if x2 is not None else 0
    cy1 = split if split is not None else cout
    y1 = torch.empty((n, h, w, cy1), dtype=x1.dtype, device=x1.device)
    y2 = torch.empty((n, h, w, cout - cy1), dtype=x1.dtype, device=x1.device) if cy1 < cout else None
    lib = _lib.load()
    need = lib.ad_conv3x3_fwd_ws_bytes(n, h, w, c1 + c2, cout, dt(x1.dtype))
    ws = _conv_workspace(x1.device, need) if need else None
    with _timed("conv3x3_fwd", 2.0 * n * h * w * 9 * (c1 + c2) * cout, float(n * h * w * (c1 + c2 + cout) * x1.element_size())):
        check(lib.ad_conv3x3_fwd(_p(x1), c1, _p(x2), c2, _p(w_packed), _p(bias), _p(y1), cy1, _p(y2),
                                 n, h, w, cout, EPI_RELU if relu else EPI_NONE, ws.ptr if ws else None,
                                 ws.nbytes if ws else 0, dt(x1.dtype), _stream()),
              "ad_conv3x3_fwd")
    return (y1, y2) if split is not None else y1


def conv3x3_dgrad_ln_bwd_is_fused(dz: torch.Tensor, cout: int) -> bool:
    n, h, w, c1 = dz.shape
    return bool(_lib.load().ad_conv3x3_dgrad_ln_bwd_is_fused(n, h, w, c1, cout, dt(dz.dtype)))


def conv3x3_dgrad_ln_bwd(dz: torch.Tensor, w_dgrad: torch.Tensor, z_prev: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor,
                         gamma: torch.Tensor, beta: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor, dbias: torch.Tensor,
                         ws: Workspace) -> torch.Tensor:
    """dgrad of a conv whose input was a Conv2D -> LayerNorm -> ReLU activation, with that layer's LayerNorm / ReLU backward
    fused: returns dz of the layer below, writes its dgamma / dbeta / dbias."""
    n, h, w, c1 = dz.shape
    cout = z_prev.shape[-1]
    out = torch.empty_like(z_prev)
    lib = _lib.load()
    ws.ensure(lib.ad_conv3x3_dgrad_ln_bwd_ws_bytes())
    with _timed("conv3x3_dgrad_ln_bwd", 2.0 * n * h * w * 9 * c1 * cout, float(n * h * w * (c1 + 2 * cout) * dz.element_size())):
        check(lib.ad_conv3x3_dgrad_ln_bwd(_p(dz), c1, _p(w_dgrad), _p(z_prev), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(out),
                                          _p(dgamma), _p(dbeta), _p(dbias), n, h, w, cout, ws.ptr, ws.nbytes, dt(dz.dtype),
                                          _stream()), "ad_conv3x3_dgrad_ln_bwd")
    return out


def conv3x3_dgrad_relu_is_fused(dz: torch.Tensor, cout: int, cy1: int) -> bool:
    n, h, w, c1 = dz.shape
    return bool(_lib.load().ad_conv3x3_dgrad_relu_is_fused(n, h, w, c1, cout, cy1, dt(dz.dtype)))


def conv3x3_dgrad_relu(dz: torch.Tensor, w_dgrad: torch.Tensor, relu_out: torch.Tensor, dbias: torch.Tensor, cout: int,
                       ws: Workspace):
    """dgrad through a conv whose first input is `relu_out` (the decoder's up-conv output) with that ReLU's gradient and
    bias gradient fused: returns (d_pre_relu [.., cy1], d_skip [.., cout - cy1] or None); writes dbias [cy1]."""
    n, h, w, c1 = dz.shape
    cy1 = relu_out.shape[-1]
    y1 = torch.empty((n, h, w, cy1), dtype=dz.dtype, device=dz.device)
    y2 = torch.empty((n, h, w, cout - cy1), dtype=dz.dtype, device=dz.device) if cy1 < cout else None
    lib = _lib.load()
    ws.ensure(lib.ad_conv3x3_dgrad_relu_ws_bytes())
    with _timed("conv3x3_dgrad_relu", 2.0 * n * h * w * 9 * c1 * cout, float(n * h * w * (c1 + cout + cy1) * dz.element_size())):
        check(lib.ad_conv3x3_dgrad_relu(_p(dz), c1, _p(w_dgrad), _p(relu_out), _p(y1), cy1, _p(y2), _p(dbias), n, h, w, cout,
                                        ws.ptr, ws.nbytes, dt(dz.dtype), _stream()), "ad_conv3x3_dgrad_relu")
    return y1, y2


def conv3x3_ln_stats_is_fused(x1: torch.Tensor, x2: Optional[torch.Tensor], cout: int) -> bool:
    """True when conv3x3_ln_relu_fwd(..., want_act=False) has a kernel for these operands."""
    n, h, w, c1 = x1.shape
    c2 = x2.shape[-1] if x2 is not None else 0
    return bool(_lib.load().ad_conv3x3_ln_stats_is_fused(n, h, w, c1, c2, cout, dt(x1.dtype)))


def conv3x3_ln_relu_fwd(x1: torch.Tensor, x2: Optional[torch.Tensor], w_packed: torch.Tensor, bias: Optional[torch.Tensor],
                        gamma: torch.Tensor, beta: torch.Tensor, cout: int, eps: float = LN_EPS, want_act: bool = True,
                        want_z: bool = True):
    """conv_block's Conv2D -> LayerNormalization -> ReLU.  Returns (z, act, mean, rstd): z is the conv output kept for
    the backward pass.  One launch where the library has the fused epilogue (cout == 64, large bf16 launches), else
    the library runs the convolution and the LayerNorm kernel back to back.
    want_act=False (conv3x3_ln_stats_is_fused must hold): the activation is not written, act is None -- for the layer in
    front of the head in a train step, whose only consumer (head_ln_bwd) re-derives it from z.
    want_z=False (inference: nothing reads z or the statistics): where the fused epilogue exists the launch writes the activation
    alone and z, mean, rstd are None; elsewhere the two launches run as always."""
    n, h, w, c1 = x1.shape
    c2 = x2.shape[-1] if x2 is not None else 0
    if (not want_z and want_act and not os.environ.get("ADUNET_LN_TWO_LAUNCHES") and os.environ.get("ADUNET_INFER_KEEP_Z") != "1"
            and _lib.load().ad_conv3x3_ln_relu_is_fused(n, h, w, c1, c2, cout, dt(x1.dtype))):
        act = torch.empty((n, h, w, cout), dtype=x1.dtype, device=x1.device)
        with _timed("conv3x3_ln_relu_fwd", 2.0 * n * h * w * 9 * (c1 + c2) * cout,        # same family, one output stream less
                    float(n * h * w * (c1 + c2 + cout) * x1.element_size())):
            check(_lib.load().ad_conv3x3_ln_relu_fwd(_p(x1), c1, _p(x2), c2, _p(w_packed), _p(bias), _p(gamma), _p(beta), eps,
                                                     None, _p(act), None, None, n, h, w, cout, None, 0, dt(x1.dtype), _stream()),
                  "ad_conv3x3_ln_relu_fwd (activation only)")
        return None, act, None, None
    if not want_act:
        z = torch.empty((n, h, w, cout), dtype=x1.dtype, device=x1.device)
        mean = torch.empty(n * h * w, dtype=torch.float32, device=x1.device)
        rstd = torch.empty(n * h * w, dtype=torch.float32, device=x1.device)
        lib = _lib.load()
        with _timed("conv3x3_ln_relu_fwd", 2.0 * n * h * w * 9 * (c1 + c2) * cout,        # same family, one output stream less
                    float(n * h * w * ((c1 + c2 + cout) * x1.element_size() + 8))):
            check(lib.ad_conv3x3_ln_relu_fwd(_p(x1), c1, _p(x2), c2, _p(w_packed), _p(bias), _p(gamma), _p(beta), eps,
                                             _p(z), None, _p(mean), _p(rstd), n, h, w, cout, None, 0, dt(x1.dtype), _stream()),
                  "ad_conv3x3_ln_relu_fwd (statistics only)")
        return z, None, mean, rstd
    # Shapes without the fused epilogue are issued as the two launches here rather than inside the library, so that
    # the per-op timers book the LayerNorm kernel under its own name (ADUNET_LN_TWO_LAUNCHES=1: A/B switch).
    if os.environ.get("ADUNET_LN_TWO_LAUNCHES") or not _lib.load().ad_conv3x3_ln_relu_is_fused(n, h, w, c1, c2, cout,
                                                                                                dt(x1.dtype)):
        z = conv3x3_fwd(x1, x2, w_packed, bias, cout)
        return (z,) + layernorm_relu_fwd(z, gamma, beta, eps=eps)
    z = torch.empty((n, h, w, cout), dtype=x1.dtype, device=x1.device)
    act = torch.empty_like(z)
    mean = torch.empty(n * h * w, dtype=torch.float32, device=x1.device)
    rstd = torch.empty(n * h * w, dtype=torch.float32, device=x1.device)
    lib = _lib.load()
    need = lib.ad_conv3x3_fwd_ws_bytes(n, h, w, c1 + c2, cout, dt(x1.dtype))
    ws = _conv_workspace(x1.device, need) if need else None
    with _timed("conv3x3_ln_relu_fwd", 2.0 * n * h * w * 9 * (c1 + c2) * cout,    # fused launches, own family
                float(n * h * w * ((c1 + c2 + 2 * cout) * x1.element_size() + 8))):
        check(lib.ad_conv3x3_ln_relu_fwd(_p(x1), c1, _p(x2), c2, _p(w_packed), _p(bias), _p(gamma), _p(beta), eps,
                                         _p(z), _p(act), _p(mean), _p(rstd), n, h, w, cout,
                                         ws.ptr if ws else None, ws.nbytes if ws else 0, dt(x1.dtype), _stream()),
              "ad_conv3x3_ln_relu_fwd")
    return z, act, mean, rstd


def conv3x3_c3_supported(x: torch.Tensor, cout: int, dtype: torch.dtype) -> bool:
    """True when the dedicated 3-input-channel first-layer kernels take this batch ([N,H,W,3] fp32, bf16 compute)."""
    if x.dim() != 4 or x.shape[-1] != 3 or x.dtype != torch.float32:
        return False
    n, h, w, _ = x.shape
    return bool(_lib.load().ad_conv3x3_c3_supported(n, h, w, cout, dt(dtype)))


def conv3x3_c3_ln_relu_fwd(x: torch.Tensor, w_hwio: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor,
                           beta: torch.Tensor, eps: float = LN_EPS, dtype: torch.dtype = torch.bfloat16, want_z: bool = True):
    """First conv_block step on the raw [N,H,W,3] fp32 input: returns (z, act, mean, rstd), z / act in `dtype` (bf16 / fp16).
    want_z=False (inference): the activation only, z / mean / rstd are None."""
    n, h, w, _ = x.shape
    if not want_z and os.environ.get("ADUNET_INFER_KEEP_Z") != "1":
        act = torch.empty((n, h, w, 64), dtype=dtype, device=x.device)
        with _timed("conv3x3_c3_ln_relu_fwd", 2.0 * n * h * w * 27 * 64, float(n * h * w * (12 + 64 * act.element_size()))):
            check(_lib.load().ad_conv3x3_c3_ln_relu_fwd(_p(x), _p(w_hwio), _p(bias), _p(gamma), _p(beta), eps, None, _p(act),
                                                        None, None, n, h, w, dt(dtype), _stream()),
                  "ad_conv3x3_c3_ln_relu_fwd (activation only)")
        return None, act, None, None
    z = torch.empty((n, h, w, 64), dtype=dtype, device=x.device)
    act = torch.empty_like(z)
    mean = torch.empty(n * h * w, dtype=torch.float32, device=x.device)
    rstd = torch.empty(n * h * w, dtype=torch.float32, device=x.device)
    with _timed("conv3x3_c3_ln_relu_fwd", 2.0 * n * h * w * 27 * 64, float(n * h * w * (12 + 2 * 64 * z.element_size() + 8))):
        check(_lib.load().ad_conv3x3_c3_ln_relu_fwd(_p(x), _p(w_hwio), _p(bias), _p(gamma), _p(beta), eps, _p(z), _p(act),
                                                    _p(mean), _p(rstd), n, h, w, dt(dtype), _stream()),
              "ad_conv3x3_c3_ln_relu_fwd")
    return z, act, mean, rstd


def conv3x3_c3_fwd(x: torch.Tensor, w_hwio: torch.Tensor, bias: Optional[torch.Tensor], dtype: torch.dtype = torch.bfloat16):
    """z = conv3x3(x) + bias on the raw [N,H,W,3] fp32 input (64 output channels), no normalisation."""
    n, h, w, _ = x.shape
    z = torch.empty((n, h, w, 64), dtype=dtype, device=x.device)
    with _timed("conv3x3_fwd", 2.0 * n * h * w * 27 * 64, float(n * h * w * (12 + 64 * z.element_size()))):
        check(_lib.load().ad_conv3x3_c3_fwd(_p(x), _p(w_hwio), _p(bias), _p(z), n, h, w, dt(dtype), _stream()), "ad_conv3x3_c3_fwd")
    return z


def conv3x3_c3_wgrad(x: torch.Tensor, dz: torch.Tensor, dw_out: torch.Tensor, ws: Workspace):
    """dw_out: fp32 [3,3,3,64] view of the flat gradient buffer; x the raw [N,H,W,3] fp32 input."""
    n, h, w, _ = x.shape
    lib = _lib.load()
    ws.ensure(lib.ad_conv3x3_c3_wgrad_ws_bytes(n, h, w))
    with _timed("conv3x3_c3_wgrad", 2.0 * n * h * w * 27 * 64, float(n * h * w * (12 + 64 * dz.element_size()))):
        check(lib.ad_conv3x3_c3_wgrad(_p(x), _p(dz), _p(dw_out), n, h, w, ws.ptr, ws.nbytes, dt(dz.dtype), _stream()),
              "ad_conv3x3_c3_wgrad")


def conv3x3_wgrad(x1: torch.Tensor, x2: Optional[torch.Tensor], dz: torch.Tensor, dw_out: torch.Tensor, cin_real: int,
                  ws: Workspace):
    """dw_out: fp32 [3,3,cin_real,cout] view (e.g. a slice of the flat gradient buffer)."""
    n, h, w, c1 = x1.shape
    c2 = x2.shape[-1] if x2 is not None else 0
    cout = dz.shape[-1]
    lib = _lib.load()
    need = lib.ad_conv3x3_wgrad_ws_bytes(n, h, w, c1 + c2, cout, dt(x1.dtype))
    ws.ensure(need)
    with _timed("conv3x3_wgrad", 2.0 * n * h * w * 9 * (c1 + c2) * cout,
                float(n * h * w * (c1 + c2 + cout) * x1.element_size())):
        check(lib.ad_conv3x3_wgrad(_p(x1), c1, _p(x2), c2, _p(dz), _p(dw_out), cin_real, n, h, w, cout,
                                   ws.ptr, ws.nbytes, dt(x1.dtype), _stream()), "ad_conv3x3_wgrad")


def layernorm_relu_fwd(z: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, relu: bool = True,
                       eps: float = LN_EPS):
    c = z.shape[-1]
    npix = z.numel() // c
    y = torch.empty_like(z)
    mean = torch.empty(npix, dtype=torch.float32, device=z.device)
    rstd = torch.empty(npix, dtype=torch.float32, device=z.device)
    with _timed("layernorm_relu_fwd"):
        check(_lib.load().ad_layernorm_relu_fwd(_p(z), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), npix, c, eps,
                                                int(relu), dt(z.dtype), _stream()), "ad_layernorm_relu_fwd")
    return y, mean, rstd


def layernorm_relu_bwd(dy, z, mean, rstd, gamma, beta, dgamma, dbeta, dbias, ws: Workspace, relu: bool = True):
    c = z.shape[-1]
    npix = z.numel() // c
    dz = torch.empty_like(z)
    lib = _lib.load()
    ws.ensure(lib.ad_layernorm_bwd_ws_bytes(npix, c))
    with _timed("layernorm_relu_bwd"):
        check(lib.ad_layernorm_relu_bwd(_p(dy), _p(z), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dz), _p(dgamma),
                                        _p(dbeta), _p(dbias), npix, c, int(relu), ws.ptr, ws.nbytes, dt(z.dtype),
                                        _stream()), "ad_layernorm_relu_bwd")
    return dz


def relu_bwd(dy, y, dbias, ws: Workspace):
    c = y.shape[-1]
    npix = y.numel() // c
    dz = torch.empty_like(y)
    lib = _lib.load()
    ws.ensure(lib.ad_layernorm_bwd_ws_bytes(npix, c))
    with _timed("relu_bwd"):
        check(lib.ad_relu_bwd(_p(dy), _p(y), _p(dz), _p(dbias), npix, c, ws.ptr, ws.nbytes, dt(y.dtype), _stream()),
              "ad_relu_bwd")
    return dz


class ResampleTables:
    """Device copies of the per-axis tap tables of one separable banded map."""

    def __init__(self, starts_y, weights_y, starts_x, weights_x, device):
        self.ky, self.kx = weights_y.shape[1], weights_x.shape[1]
        self.oh, self.ow = weights_y.shape[0], weights_x.shape[0]
        self.sy = torch.tensor(np.ascontiguousarray(starts_y, dtype=np.int32), device=device)
        self.sx = torch.tensor(np.ascontiguousarray(starts_x, dtype=np.int32), device=device)
        self.wy = torch.tensor(np.ascontiguousarray(weights_y, dtype=np.float32), device=device)
        self.wx = torch.tensor(np.ascontiguousarray(weights_x, dtype=np.float32), device=device)


def resample(x: torch.Tensor, tab: ResampleTables, out: Optional[torch.Tensor] = None, accumulate: bool = False):
    n, h, w, c = x.shape
    if out is None:
        assert not accumulate
        out = torch.empty((n, tab.oh, tab.ow, c), dtype=x.dtype, device=x.device)
    with _timed("resample", 0.0, float((x.numel() + out.numel() * (2 if accumulate else 1)) * x.element_size())):
        check(_lib.load().ad_resample(_p(x), _p(out), _p(tab.sy), _p(tab.wy), tab.ky, _p(tab.sx), _p(tab.wx), tab.kx,
                                      n, h, w, tab.oh, tab.ow, c, int(accumulate), dt(x.dtype), _stream()), "ad_resample")
    return out


# ---- the decoder step "dec_up -> Conv2D(nf, 3, same, relu)" without the up-resized tensor (csrc/upconv.hip)
class UpconvTables:
    """Device tables of one up-resize (h, w) -> (oh, ow) for the gather kernels: the two-tap forward form and the
    transposed spans.  `ok` is False when the resize is not an up-resize with at most two taps per output index or the
    transposed horizontal span is wider than the backward kernel's register window."""

    def __init__(self, h: int, w: int, oh: int, ow: int, device):
        from . import resize_tables as rt
        fy, fx = rt.up_taps2(h, oh), rt.up_taps2(w, ow)
        self.ok = fy is not None and fx is not None
        # the forward kernel's four-row window needs non-decreasing starts that advance by at most 2 over two output rows
        self.ok = self.ok and all(bool((np.diff(t[0]) >= 0).all()) and (len(t[0]) < 3 or int((t[0][2:] - t[0][:-2]).max()) <= 2)
                                  for t in (fy, fx))
        if not self.ok:
            return
        self.window = 3 if len(fy[0]) < 3 or int((fy[0][2:] - fy[0][:-2]).max()) <= 1 else 4
        self._sx_host, self._slab_cols = np.asarray(fx[0], np.int64), {}
        ty, tx = rt.aa_spans_transposed(h, oh), rt.aa_spans_transposed(w, ow)
        self.kyt, self.kxt = ty[1].shape[1], tx[1].shape[1]
        self.ok = bool(_lib.load().ad_upconv_gather_bwd_supported(self.kxt)) and self.kyt <= 30
        self.h, self.w, self.oh, self.ow = h, w, oh, ow
        dev_i = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.int32), device=device)
        dev_f = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device=device)
        self.sy, self.wy, self.sx, self.wx = dev_i(fy[0]), dev_f(fy[1]), dev_i(fx[0]), dev_f(fx[1])
        self.ryt, self.wyt, self.cxt, self.wxt = dev_i(ty[0]), dev_f(ty[1]), dev_i(tx[0]), dev_f(tx[1])


    def slab_cols(self, c: int, dtype: torch.dtype) -> int:
        """Most low-resolution columns the output columns of one workgroup of ad_upconv_gather_fwd (and their +-1 neighbours)
        read: the piece of a bank row it stages in LDS.  Computed by the library from the host copy of the table the launch
        passes (ad_upconv_slab_cols), so the staging rule lives in one place; -1 = table / channel count not acceptable."""
        key = (c, dtype)
        if key not in self._slab_cols:
            sx = np.ascontiguousarray(self._sx_host, dtype=np.int32)
            self._slab_cols[key] = int(_lib.load().ad_upconv_slab_cols(sx.ctypes.data, self.w, self.ow, c, dt(dtype)))
        return self._slab_cols[key]

    def gather_fwd_ok(self, c: int, dtype: torch.dtype) -> bool:
        cols = self.slab_cols(c, dtype)
        return cols > 0 and bool(_lib.load().ad_upconv_gather_fwd_supported(c, cols, dt(dtype)))


def pw_supported(m: int, k: int, n: int, dtype: torch.dtype) -> bool:
    return bool(_lib.load().ad_pw_supported(m, k, n, dt(dtype)))


def pw_bank_pack(w_hwio: torch.Tensor, dtype: torch.dtype, out=None):
    """fp32 Keras kernel [3,3,cin,cout] -> (bank_fwd, bank_bwd) GEMM operands; `out` = a previous result refreshed in place."""
    kh, kw, cin, cout = w_hwio.shape
    assert (kh, kw) == (3, 3) and w_hwio.dtype == torch.float32
    lib = _lib.load()
    if out is not None:
        bf, bd = out
    else:
        ne = lib.ad_pw_bank_elems(cin, cout)
        bf = torch.empty(ne, dtype=dtype, device=w_hwio.device)
        bd = torch.empty(ne, dtype=dtype, device=w_hwio.device)
    with _timed("pw_bank_pack"):
        check(lib.ad_pw_bank_pack(_p(w_hwio), cin, cout, _p(bf), _p(bd), dt(dtype), _stream()), "ad_pw_bank_pack")
    return bf, bd


def pw_gemm(x: torch.Tensor, bank: torch.Tensor, n_out: int) -> torch.Tensor:
    """y[..., n_out] = x[..., k] @ bank[k, n_out] over the flattened pixels of an NHWC tensor."""
    k = x.shape[-1]
    m = x.numel() // k
    y = torch.empty(x.shape[:-1] + (n_out,), dtype=x.dtype, device=x.device)
    with _timed("pw_gemm", 2.0 * m * k * n_out, float(m * (k + n_out) * x.element_size())):
        check(_lib.load().ad_pw_gemm(_p(x), _p(bank), _p(y), m, k, n_out, dt(x.dtype), _stream()), "ad_pw_gemm")
    return y


def upconv_gather_fwd(ybank: torch.Tensor, bias: Optional[torch.Tensor], tab: UpconvTables, relu: bool = True) -> torch.Tensor:
    n, h, w, c9 = ybank.shape
    c = c9 // 9
    assert (h, w) == (tab.h, tab.w)
    out = torch.empty((n, tab.oh, tab.ow, c), dtype=ybank.dtype, device=ybank.device)
    ept = 4 if ybank.dtype != torch.float32 else 2
    if c % ept or 256 % (c // ept):
        raise ValueError(f"upconv_gather_fwd: {c} channels (a divisor of {256 * ept} in steps of {ept} is needed)")
    with _timed("upconv_gather_fwd", 0.0, float((ybank.numel() + out.numel()) * ybank.element_size())):
        check(_lib.load().ad_upconv_gather_fwd(_p(ybank), _p(bias), _p(out), _p(tab.sy), _p(tab.wy), _p(tab.sx), _p(tab.wx),
                                               tab.window, tab.slab_cols(c, ybank.dtype), n, h, w, tab.oh, tab.ow, c, int(relu), dt(ybank.dtype), _stream()),
              "ad_upconv_gather_fwd")
    return out


def upconv_gather_bwd(g: torch.Tensor, tab: UpconvTables) -> torch.Tensor:
    n, oh, ow, c = g.shape
    assert (oh, ow) == (tab.oh, tab.ow)
    dyb = torch.empty((n, tab.h, tab.w, 9 * c), dtype=g.dtype, device=g.device)
    with _timed("upconv_gather_bwd", 0.0, float((g.numel() + dyb.numel()) * g.element_size())):
        check(_lib.load().ad_upconv_gather_bwd(_p(g), _p(dyb), _p(tab.ryt), _p(tab.wyt), tab.kyt, _p(tab.cxt), _p(tab.wxt),
                                               tab.kxt, n, tab.h, tab.w, oh, ow, c, dt(g.dtype), _stream()),
              "ad_upconv_gather_bwd")
    return dyb


def upconv_bank_wgrad(x_low: torch.Tensor, dybank: torch.Tensor, dw_out: torch.Tensor, ws: Workspace):
    """dW [3,3,cin,cout] (fp32 view of the flat gradient buffer) = re-ordered x_low^T dybank (contraction over pixels)."""
    cin, n9 = x_low.shape[-1], dybank.shape[-1]
    m = x_low.numel() // cin
    lib = _lib.load()
    if lib.ad_pw_wgrad_supported(m, cin, n9 // 9, dt(x_low.dtype)):         # one pass over x and dY (16-bit types)
        ws.ensure(lib.ad_pw_wgrad_ws_bytes(m, cin, n9 // 9))
        with _timed("pw_wgrad", 2.0 * m * cin * n9, float(m * (cin + n9) * x_low.element_size())):
            check(lib.ad_pw_wgrad(_p(x_low), _p(dybank), _p(dw_out), m, cin, n9 // 9, ws.ptr, ws.nbytes, dt(x_low.dtype), _stream()),
                  "ad_pw_wgrad")
        return
    dw9 = torch.empty((3, 3, cin, n9), dtype=torch.float32, device=x_low.device)
    conv3x3_wgrad(x_low.view(m, 1, 1, cin), None, dybank.view(m, 1, 1, n9), dw9, cin, ws)
    with _timed("pw_bank_grad"):
        check(_lib.load().ad_pw_bank_grad(_p(dw9), cin, n9 // 9, _p(dw_out), _stream()), "ad_pw_bank_grad")


def resample_ln_bwd_supported(dskip: torch.Tensor, tab: ResampleTables) -> bool:
    n, oh, ow, c = dskip.shape
    return bool(_lib.load().ad_resample_ln_bwd_supported(n, oh, ow, c, tab.kx, dt(dskip.dtype)))


def resample_ln_bwd(d_low, tab: ResampleTables, dskip, z, mean, rstd, gamma, beta, dgamma, dbeta, dbias, ws: Workspace):
    """dz of the conv_block that produced a skip tensor: LayerNorm/ReLU backward of (dskip + resample^T(d_low)) in one
    pass (the sum is never stored).  `tab`: the transposed tables of the down-resize."""
    n, h, w, c = d_low.shape
    _, oh, ow, _ = dskip.shape
    dz = torch.empty_like(z)
    lib = _lib.load()
    ws.ensure(lib.ad_resample_ln_bwd_ws_bytes(n, oh, ow, c, dt(z.dtype)))
    with _timed("resample_ln_bwd", 0.0, float((d_low.numel() + 3 * z.numel()) * z.element_size())):
        check(lib.ad_resample_ln_bwd(_p(d_low), _p(dskip), _p(z), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dz), _p(dgamma),
                                     _p(dbeta), _p(dbias), _p(tab.sy), _p(tab.wy), tab.ky, _p(tab.sx), _p(tab.wx), tab.kx,
                                     n, h, w, oh, ow, c, ws.ptr, ws.nbytes, dt(z.dtype), _stream()), "ad_resample_ln_bwd")
    return dz


def head_fwd(xh, w, b, inp, target, ws: Workspace, loss_kind: int = 0, eps: float = CHARBONNIER_EPS):
    """Returns (out[n,h,w,3] fp32, stats[3] = (loss SUM, mean PSNR, loss MEAN) or None, sqerr[n] or None)."""
    n, h, wd, ch = xh.shape
    out = torch.empty((n, h, wd, 3), dtype=torch.float32, device=xh.device)
    stats = sqerr = None
    if target is not None:
        stats = torch.empty(3, dtype=torch.float32, device=xh.device)
        sqerr = torch.empty(n, dtype=torch.float32, device=xh.device)
    lib = _lib.load()
    ws.ensure(lib.ad_head_ws_bytes(n, ch))
    with _timed("head_fwd", 0.0, float(xh.numel() * xh.element_size() + (3 if target is not None else 2) * out.numel() * 4)):
        check(lib.ad_head_fwd(_p(xh), _p(w), _p(b), _p(inp), _p(target), _p(out), _p(stats), _p(sqerr), n, h * wd, ch,
                              loss_kind, eps, ws.ptr, ws.nbytes, dt(xh.dtype), _stream()), "ad_head_fwd")
    return out, stats, sqerr


def head_bwd(xh, w, b, inp, target, dw, db, grad_scale: float, ws: Workspace, loss_kind: int = 0,
             eps: float = CHARBONNIER_EPS, loss_scale: Optional[torch.Tensor] = None):
    """loss_scale: the device-resident scaler state of a LossScaleOptimizer (its first float multiplies the gradient)."""
    n, h, wd, ch = xh.shape
    dxh = torch.empty_like(xh)
    lib = _lib.load()
    ws.ensure(lib.ad_head_ws_bytes(n, ch))
    with _timed("head_bwd"):
        check(lib.ad_head_bwd(_p(xh), _p(w), _p(b), _p(inp), _p(target), _p(dxh), _p(dw), _p(db), n, h * wd, ch, loss_kind,
                              eps, grad_scale, _p(loss_scale), ws.ptr, ws.nbytes, dt(xh.dtype), _stream()), "ad_head_bwd")
    return dxh


def head_ln_bwd(xh, w, b, inp, target, z, mean, rstd, gamma, beta, dw, db, dgamma, dbeta, dbias_conv, grad_scale: float,
                ws: Workspace, loss_kind: int = 0, eps: float = CHARBONNIER_EPS, loss_scale: Optional[torch.Tensor] = None,
                stats: Optional[torch.Tensor] = None, sqerr: Optional[torch.Tensor] = None):
    """head_bwd + the LayerNorm/ReLU backward of the layer feeding the head, one pass; returns that layer's dz.
    stats[3] / sqerr[n]: filled with what head_fwd reports (loss sum, mean PSNR, loss mean / per-image squared error).
    xh=None: the head's input is re-derived from z (the forward pass did not store it: conv3x3_ln_relu_fwd, want_act=False)."""
    n, h, wd, ch = z.shape
    dz = torch.empty_like(z)
    lib = _lib.load()
    ws.ensure(lib.ad_head_ln_bwd_ws_bytes(n, ch))
    with _timed("head_ln_bwd", 0.0, float((3 if xh is not None else 2) * z.numel() * z.element_size() + 2 * inp.numel() * 4)):
        check(lib.ad_head_ln_bwd(_p(xh), _p(w), _p(b), _p(inp), _p(target), _p(z), _p(mean), _p(rstd), _p(gamma), _p(beta),
                                 _p(dz), _p(dw), _p(db), _p(dgamma), _p(dbeta), _p(dbias_conv), n, h * wd, ch, loss_kind, eps,
                                 grad_scale, _p(loss_scale), _p(stats), _p(sqerr), ws.ptr, ws.nbytes, dt(z.dtype), _stream()),
              "ad_head_ln_bwd")
    return dz


def adam_step(p, g, m, v, step: int, lr=1e-4, b1=0.9, b2=0.999, eps=1e-7, gscale: float = 1.0):
    with _timed("adam_step", 0.0, 28.0 * p.numel()):
        check(_lib.load().ad_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, b1, b2, eps, step, gscale, _stream()),
              "ad_adam_step")


def adam_alpha(lr: float, b1: float, b2: float, step: int) -> float:
    return float(_lib.load().ad_adam_alpha(lr, b1, b2, step))


def adam_step_dev(p, g, m, v, alpha_dev: torch.Tensor, b1=0.9, b2=0.999, eps=1e-7, gscale: float = 1.0):
    """Adam update whose step-dependent factor is read from device memory (hipGraph-replayable)."""
    with _timed("adam_step", 0.0, 28.0 * p.numel()):
        check(_lib.load().ad_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(alpha_dev), b1, b2, eps, gscale,
                                           _stream()), "ad_adam_step_dev")


def loss_scale_check(g: torch.Tensor, state: torch.Tensor):
    """Sets state[3] when the flat gradient buffer holds an inf / NaN (Keras LossScaleOptimizer's finiteness test)."""
    with _timed("loss_scale"):
        check(_lib.load().ad_loss_scale_check(_p(g), g.numel(), _p(state), _stream()), "ad_loss_scale_check")


def loss_scale_update(state: torch.Tensor, growth_steps: int):
    with _timed("loss_scale"):
        check(_lib.load().ad_loss_scale_update(_p(state), growth_steps, _stream()), "ad_loss_scale_update")


def adam_step_scaled(p, g, m, v, lr_dev: torch.Tensor, state: torch.Tensor, b1=0.9, b2=0.999, eps=1e-7, gscale: float = 1.0):
    """Adam under a device-resident loss scaler: skipped on overflow, gradients unscaled, applied-step bias correction."""
    with _timed("adam_step", 0.0, 28.0 * p.numel()):
        check(_lib.load().ad_adam_step_scaled(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(lr_dev), b1, b2, eps, gscale,
                                              _p(state), _stream()), "ad_adam_step_scaled")


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(_lib.load().ad_cast(_p(x), dt(x.dtype), _p(y), dt(dtype), x.numel(), _stream()), "ad_cast")
    return y


# --------------------------------------------------------------------------- tier 2 (segmentation models)
BN_EPS, BN_MOMENTUM = 1e-3, 0.99   # Keras BatchNormalization defaults (Segmenation/code/train_adaptive_unet.py:327)


def batchnorm_relu_fwd_train(z, gamma, beta, moving_mean, moving_var, ws: Workspace, relu: bool = True,
                             eps: float = BN_EPS, momentum: float = BN_MOMENTUM):
    """Returns (y, save_mean, save_rstd); updates moving_mean / moving_var in place (may be None)."""
    c = z.shape[-1]
    npix = z.numel() // c
    y = torch.empty_like(z)
    mean, rstd, var = (torch.empty(c, dtype=torch.float32, device=z.device) for _ in range(3))
    lib = _lib.load()
    ws.ensure(lib.ad_batchnorm_ws_bytes(c))
    with _timed("batchnorm_fwd", 0.0, 2.0 * z.numel() * z.element_size()):       # algorithmic: read z once, write y
        check(lib.ad_batchnorm_relu_fwd_train(_p(z), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _p(var),
                                              _p(moving_mean), _p(moving_var), momentum, npix, c, eps, int(relu),
                                              ws.ptr, ws.nbytes, dt(z.dtype), _stream()), "ad_batchnorm_relu_fwd_train")
    return y, mean, rstd


def batchnorm_pool_supported(z) -> bool:
    """batchnorm_relu_pool_fwd_train takes z: even extents, at most 256 16-byte channel vectors per pixel."""
    n, h, w, c = z.shape
    ept = 16 // z.element_size()
    return h % 2 == 0 and w % 2 == 0 and h >= 2 and w >= 2 and c % ept == 0 and c // ept <= 256 and 256 % (c // ept) == 0


def batchnorm_relu_pool_fwd_train(z, gamma, beta, moving_mean, moving_var, ws: Workspace, relu: bool = True,
                                  eps: float = BN_EPS, momentum: float = BN_MOMENTUM):
    """batchnorm_relu_fwd_train whose pass over z also writes MaxPooling2D(2) of the activation: returns
    (y, pooled, save_mean, save_rstd)."""
    n, h, w, c = z.shape
    y = torch.empty_like(z)
    pooled = torch.empty((n, h // 2, w // 2, c), dtype=z.dtype, device=z.device)
    mean, rstd, var = (torch.empty(c, dtype=torch.float32, device=z.device) for _ in range(3))
    lib = _lib.load()
    ws.ensure(lib.ad_batchnorm_ws_bytes(c))
    with _timed("batchnorm_fwd", 0.0, 2.25 * z.numel() * z.element_size()):      # read z once, write y and its pooling
        check(lib.ad_batchnorm_relu_pool_fwd_train(_p(z), _p(gamma), _p(beta), _p(y), _p(pooled), _p(mean), _p(rstd), _p(var),
                                                   _p(moving_mean), _p(moving_var), momentum, n, h, w, c, eps, int(relu),
                                                   ws.ptr, ws.nbytes, dt(z.dtype), _stream()), "ad_batchnorm_relu_pool_fwd_train")
    return y, pooled, mean, rstd


def batchnorm_relu_fwd_infer(z, gamma, beta, moving_mean, moving_var, relu: bool = True, eps: float = BN_EPS):
    c = z.shape[-1]
    y = torch.empty_like(z)
    tmp = torch.empty(c, dtype=torch.float32, device=z.device)
    with _timed("batchnorm_fwd", 0.0, 2.0 * z.numel() * z.element_size()):
        check(_lib.load().ad_batchnorm_relu_fwd_infer(_p(z), _p(gamma), _p(beta), _p(moving_mean), _p(moving_var), _p(y),
                                                      _p(tmp), z.numel() // c, c, eps, int(relu), dt(z.dtype), _stream()),
              "ad_batchnorm_relu_fwd_infer")
    return y


def batchnorm_relu_bwd(dy, z, mean, rstd, gamma, beta, dgamma, dbeta, ws: Workspace, relu: bool = True, dbias=None):
    """dbias (fp32 [c], optional): receives the column sums of dz as stored -- the bias gradient of the convolution in
    front -- from the pass that writes dz (no separate colsum launch)."""
    c = z.shape[-1]
    dz = torch.empty_like(z)
    lib = _lib.load()
    ws.ensure(lib.ad_batchnorm_ws_bytes(c))
    with _timed("batchnorm_bwd", 0.0, 3.0 * z.numel() * z.element_size()):       # algorithmic: read dy and z, write dz
        check(lib.ad_batchnorm_relu_bwd_dbias(_p(dy), _p(z), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dz), _p(dgamma),
                                              _p(dbeta), _p(dbias), z.numel() // c, c, int(relu), ws.ptr, ws.nbytes,
                                              dt(z.dtype), _stream()), "ad_batchnorm_relu_bwd")
    return dz


def colsum(x, out, ws: Workspace):
    c = x.shape[-1]
    lib = _lib.load()
    ws.ensure(lib.ad_batchnorm_ws_bytes(c))
    with _timed("colsum", 0.0, float(x.numel() * x.element_size())):
        check(lib.ad_colsum(_p(x), _p(out), x.numel() // c, c, ws.ptr, ws.nbytes, dt(x.dtype), _stream()), "ad_colsum")
    return out


def maxpool2_fwd(x):
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    with _timed("maxpool2", 0.0, 1.25 * x.numel() * x.element_size()):
        check(_lib.load().ad_maxpool2_fwd(_p(x), _p(y), n, h, w, c, dt(x.dtype), _stream()), "ad_maxpool2_fwd")
    return y


def maxpool2_bwd(dy, x, add=None):
    """add ([n,h,w,c], optional): summed into the result in the same pass (the skip connection's gradient at an encoder junction)."""
    n, h, w, c = x.shape
    dx = torch.empty_like(x)
    with _timed("maxpool2", 0.0, (2.25 + (add is not None)) * x.numel() * x.element_size()):
        check(_lib.load().ad_maxpool2_bwd_add(_p(dy), _p(x), _p(add), _p(dx), n, h, w, c, dt(x.dtype), _stream()), "ad_maxpool2_bwd")
    return dx


def pixel_shuffle2(x, to_space: bool):
    """to_space: [n,h,w,4c] -> [n,2h,2w,c]; else [n,2h,2w,c] -> [n,h,w,4c]."""
    if to_space:
        n, h, w, c4 = x.shape
        c = c4 // 4
        y = torch.empty((n, 2 * h, 2 * w, c), dtype=x.dtype, device=x.device)
    else:
        n, h2, w2, c = x.shape
        h, w = h2 // 2, w2 // 2
        y = torch.empty((n, h, w, 4 * c), dtype=x.dtype, device=x.device)
    with _timed("pixel_shuffle2"):
        check(_lib.load().ad_pixel_shuffle2(_p(x), _p(y), n, h, w, c, int(to_space), dt(x.dtype), _stream()),
              "ad_pixel_shuffle2")
    return y


def conv_transpose2x2s2_pack(w_t: torch.Tensor, dtype: torch.dtype, out=None):
    """Keras Conv2DTranspose kernel [2,2,Cout,Cin] -> operand packs of the equivalent pointwise GEMM Cin -> 4*Cout
    (output block a*2+b holds W[a,b]^T).  Returns (w_fwd, w_dgrad, staging).  `out` = a previous result: the packs and
    the fp32 staging tensor are refreshed in place, so they keep their addresses (a captured hipGraph relies on it)."""
    kh, kw, cout, cin = w_t.shape
    assert (kh, kw) == (2, 2)
    wg = w_t.permute(3, 0, 1, 2).reshape(cin, 4 * cout)               # [Cin][(a,b,o)]
    if out is not None:
        wf, wd, w9 = out
    else:
        wf = wd = None
        w9 = torch.zeros((3, 3, cin, 4 * cout), dtype=torch.float32, device=w_t.device)
    w9[1, 1].copy_(wg)                                                # pointwise = centre tap of the 3x3 operand
    wf, wd = conv3x3_pack(w9, cin, dtype, want_dgrad=True, out=(wf, wd) if wf is not None else None)
    return wf, wd, w9


def conv_transpose2x2s2_fwd(x, w_fwd, bias, cout: int):
    """y[n,2h,2w,cout] = Conv2DTranspose(cout, 2, strides=2)(x): pointwise GEMM on the matrix cores + depth-to-space."""
    n, h, w, cin = x.shape
    b4 = bias.repeat(4).contiguous() if bias is not None else None
    flat = conv3x3_fwd(x.view(n * h * w, 1, 1, cin), None, w_fwd, b4, 4 * cout)
    return pixel_shuffle2(flat.view(n, h, w, 4 * cout), to_space=True)


def conv_transpose2x2s2_bwd(x, dy, w_dgrad, dw_t: torch.Tensor, db: torch.Tensor, ws: Workspace):
    """Returns dx; writes dw_t [2,2,Cout,Cin] (Keras layout) and db [Cout]."""
    n, h, w, cin = x.shape
    cout = dy.shape[-1]
    g = pixel_shuffle2(dy, to_space=False).view(n * h * w, 1, 1, 4 * cout)
    dx = conv3x3_fwd(g, None, w_dgrad, None, cin).view(n, h, w, cin)
    dw9 = torch.empty((3, 3, cin, 4 * cout), dtype=torch.float32, device=x.device)
    conv3x3_wgrad(x.view(n * h * w, 1, 1, cin), None, g, dw9, cin, ws)
    dw_t.copy_(dw9[1, 1].view(cin, 2, 2, cout).permute(1, 2, 3, 0))
    colsum(g.view(-1, cout), db, ws)        # bias grad: sum over pixels and the 4 sub-positions
    return dx


def seg_head_fwd(xh, w, b, target, ws: Workspace, counts: bool = False):
    """Returns (prob [n,h,w,1] fp32, sums [n,3] or None); counts=True: sums is [n,9], columns 3.. = the thresholded counts
    and unclipped dice sums of ad_seg_head_fwd_counts (the vanilla baseline's Keras metrics)."""
    n, h, wd, ch = xh.shape
    prob = torch.empty((n, h, wd, 1), dtype=torch.float32, device=xh.device)
    sums = torch.empty((n, 3), dtype=torch.float32, device=xh.device) if target is not None else None
    cnt = torch.empty((n, 6), dtype=torch.float32, device=xh.device) if (counts and target is not None) else None
    lib = _lib.load()
    ws.ensure(lib.ad_seg_head_ws_bytes(n, ch))
    with _timed("seg_head_fwd", 0.0, float(xh.numel() * xh.element_size() + n * h * wd * (8 if target is not None else 4))):
        check(lib.ad_seg_head_fwd_counts(_p(xh), _p(w), _p(b), _p(target), _p(prob), _p(sums), _p(cnt), n, h * wd, ch, ws.ptr,
                                         ws.nbytes, dt(xh.dtype), _stream()), "ad_seg_head_fwd")
    if cnt is not None:
        return prob, (sums, cnt)
    return prob, sums


def seg_metrics(sums, count: float, bce_weight: float, dice_weight: float, smooth: float = 1e-6):
    """(loss, dice, iou) of a batch from seg_head_fwd's per-sample sums, as a device tensor [3] (one launch)."""
    out = torch.empty(3, dtype=torch.float32, device=sums.device)
    with _timed("seg_metrics"):
        check(_lib.load().ad_seg_metrics(_p(sums), sums.shape[0], float(count), float(bce_weight), float(dice_weight),
                                         float(smooth), _p(out), _stream()), "ad_seg_metrics")
    return out


def softmax_head_fwd(xh, w, b):
    """Conv2D(K, 1, softmax) head: xh [n,h,w,ch], w [ch,K] fp32, b [K] -> probabilities [n,h,w,K] fp32 (forward only)."""
    n, h, wd, ch = xh.shape
    k = w.shape[-1]
    prob = torch.empty((n, h, wd, k), dtype=torch.float32, device=xh.device)
    with _timed("softmax_head_fwd"):
        check(_lib.load().ad_softmax_head_fwd(_p(xh), _p(w), _p(b), _p(prob), n * h * wd, ch, k, dt(xh.dtype), _stream()),
              "ad_softmax_head_fwd")
    return prob


def seg_head_bwd(xh, w, target, prob, sums, dw, db, bce_weight: float, dice_weight: float, ws: Workspace,
                 smooth: float = 1e-6, loss_scale: Optional[torch.Tensor] = None):
    n, h, wd, ch = xh.shape
    dxh = torch.empty_like(xh)
    lib = _lib.load()
    ws.ensure(lib.ad_seg_head_ws_bytes(n, ch))
    with _timed("seg_head_bwd", 0.0, float(2 * xh.numel() * xh.element_size() + n * h * wd * 8)):
        check(lib.ad_seg_head_bwd(_p(xh), _p(w), _p(target), _p(prob), _p(sums), _p(dxh), _p(dw), _p(db), n, h * wd, ch,
                                  bce_weight, dice_weight, smooth, _p(loss_scale), ws.ptr, ws.nbytes, dt(xh.dtype), _stream()),
              "ad_seg_head_bwd")
    return dxh
