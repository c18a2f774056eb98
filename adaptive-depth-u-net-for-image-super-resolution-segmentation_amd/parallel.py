"""Single-node data parallelism: one process per GPU, RCCL all-reduce of the flat gradient buffer over
xGMI, bucketed and overlapped with the backward pass.

The reference is strictly single-GPU, so the only contract is mathematical: an N-rank
step with per-rank batch b equals a one-rank step with batch N*b (LayerNorm has no cross-sample statistic).
Each rank scales its loss gradient by 1/(local elements); the summed gradients are then multiplied by
1/world_size inside the Adam kernel (`gscale`), which yields the global-batch mean.

Buckets are contiguous ranges of the flat gradient buffer cut at parameter boundaries, ordered from the END
of the buffer: backward produces gradients in reverse creation order, so the tail (head + decoder, the
largest early-finishing tensors) is reduced first on a dedicated communication stream while the encoder's
wgrads are still running.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): few, large (default 32 MiB)
messages keep every link busy without paying the per-collective latency 20+ times per step.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def plan_buckets(index, total: int, bucket_elems: int) -> List[Tuple[int, int]]:
    """Cut [0, total) at parameter offsets into ranges of >= bucket_elems, listed from the end backwards."""
    offsets = sorted({off for off, _ in index.values()})
    buckets: List[Tuple[int, int]] = []
    hi = total
    for off in reversed(offsets):
        if hi - off >= bucket_elems or off == 0:
            if hi > off:
                buckets.append((off, hi))
            hi = off
    if hi > 0:
        buckets.append((0, hi))
    return buckets


class _EventWork:
    """`wait()` of a native all-reduce: the current (compute) stream waits for the event recorded after it."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class DataParallel:
    """Attach to a Model: broadcasts rank-0 weights, then all-reduces gradients bucket by bucket."""

    def __init__(self, model, bucket_bytes: int = 32 << 20, group=None, native: Optional[bool] = None):
        """native: run the bucket all-reduces through the library's own RCCL entry point (`ad_allreduce_bucket`,
        include/adunet.h) instead of `torch.distributed.all_reduce`; default from ADUNET_NATIVE_RCCL=1.  torch.distributed
        is then only the bootstrap channel for the 128-byte RCCL unique id and the initial weight broadcast."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group)
        self._comm = None
        self._native = None
        if native is None:
            native = os.environ.get("ADUNET_NATIVE_RCCL") == "1"
        if native and model.G.is_cuda:
            self._native = self._create_native_comm(model.G.device, group)
        self.buckets = plan_buckets(model.index, model.count_params(), max(1, bucket_bytes // 4))
        self._next = 0
        self._works = []
        self._waits = []
        self.measure = False          # bench.py: time the compute stream's wait for the exchange with HIP events
        self._cuda = model.G.is_cuda
        self._comm = torch.cuda.Stream(device=model.G.device) if self._cuda else None
        dist.broadcast(model.P, src=0, group=group)
        if hasattr(model, "_repack"):
            model._repack()
        model.grad_ready = self._ready
        model.grad_sync = self._sync
        model._dp = self

    def _create_native_comm(self, device, group):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        uid = torch.zeros(128, dtype=torch.uint8)
        if dist.get_rank(group) == 0:
            buf = (C.c_char * 128)()
            _lib.check(lib.ad_comm_unique_id(buf), "ad_comm_unique_id")
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        uid = uid.to(device)
        dist.broadcast(uid, src=0, group=group)
        raw = bytes(uid.cpu().tolist())
        comm = C.c_void_p()
        # ncclCommInitRank binds the communicator to the CURRENT HIP device: make that the model's device, whatever the
        # caller's current device is
        with torch.cuda.device(device):
            torch.cuda.synchronize(device)
            _lib.check(lib.ad_comm_create(raw, dist.get_rank(group), self.world, C.byref(comm)), "ad_comm_create")
        self._lib, self._check = lib, _lib.check
        return comm

    def __del__(self):          # error paths that never reach close(): do not leak the RCCL communicator
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        """Destroy the native communicator (after the last step, before the process group goes away)."""
        if getattr(self, "_native", None) is not None:
            comm, self._native = self._native, None
            with torch.cuda.device(self.model.G.device):
                torch.cuda.synchronize()
                self._check(self._lib.ad_comm_destroy(comm), "ad_comm_destroy")

    def _launch(self, lo: int, hi: int):
        g = self.model.G[lo:hi]
        if self._native is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._comm.wait_event(ev)
            with torch.cuda.device(g.device):
                self._check(self._lib.ad_allreduce_bucket(self._native, g.data_ptr(), hi - lo, self._comm.cuda_stream),
                            "ad_allreduce_bucket")
            done = torch.cuda.Event()
            done.record(self._comm)
            self._works.append(_EventWork(done))
            return
        if self._cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm):
                self._comm.wait_event(ev)
                self._works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _ready(self, low_offset: int):
        """Every gradient at flat offset >= low_offset is final."""
        while self._next < len(self.buckets) and self.buckets[self._next][0] >= low_offset:
            self._launch(*self.buckets[self._next])
            self._next += 1

    def _sync(self, model) -> float:
        while self._next < len(self.buckets):
            self._launch(*self.buckets[self._next])
            self._next += 1
        self.wait_all()
        self._next = 0
        return 1.0 / self.world

    def wait_all(self):
        timed = self.measure and self._cuda and self._works
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._works:
            w.wait()           # the compute stream waits for the collective; the host does not block
        if timed:              # e0 -> e1 on the compute stream = the time it sat idle waiting for the exchange
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._waits.append((e0, e1))
        self._works.clear()

    def exposed_ms(self) -> float:
        """Total time the compute stream waited for gradient exchange since `measure` was switched on (call after a
        device synchronize); clears the record."""
        total = sum(a.elapsed_time(b) for a, b in self._waits)
        self._waits.clear()
        return total
