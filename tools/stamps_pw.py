#!/usr/bin/env python3
"""Diagnostic (tools/build_variant.sh stamp -DAD_STAMP; ADUNET_LIB=ab/stamp.so): where wave 0 of every workgroup of the LDS-tiled
bank GEMM (pw_gemm_lds_kernel) spends its cycles.  Stamps serialise what the kernel overlaps: read the shares."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adunet_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
fn = lib.ad_dbg_set_pw_stamp_buffer
fn.argtypes = [ctypes.c_void_p]
for (m, k, n) in [(32 * 34 * 34, 1024, 4608), (32 * 34 * 34, 4608, 1024), (32 * 56 * 56, 512, 2304)]:
    g = torch.Generator().manual_seed(1)
    x = (torch.rand((1, 1, m, k), generator=g) - 0.5).to(device=dev, dtype=torch.bfloat16)
    bank = ((torch.rand((k * n,), generator=g) - 0.5) * 0.05).to(device=dev, dtype=torch.bfloat16)
    for _ in range(20):
        ops.pw_gemm(x, bank, n)
    dbg = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
    fn(dbg.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.pw_gemm(x, bank, n); e1.record()
    torch.cuda.synchronize()
    fn(None)
    d = dbg.view(256, 8).double().cpu()
    d = d[d[:, 7] > 0]
    tot = d[:, 7].mean()
    names = ["wait for staged loads + LDS stores", "barrier", "issue of the next loads", "fragment reads + MFMAs", "tile epilogue"]
    print(f"m={m} k={k} n={n}: variant {lib.ad_pw_gemm_variant(m, k, n, ops.dt(torch.bfloat16))}, {e0.elapsed_time(e1) * 1e3:.0f} us (stamped build), "
          f"{2.0 * m * k * n / e0.elapsed_time(e1) / 1e9:.0f} TFLOP/s")
    for i, nm in enumerate(names):
        print(f"   {nm:<38} {d[:, i].mean() / tot:6.1%}")
