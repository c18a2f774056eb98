#!/usr/bin/env python3
"""Diagnostic (build with tools/build_variant.sh stamp -DAD_STAMP; run with ADUNET_LIB=ab/stamp.so): where MFMA wave 0 of every
workgroup of the fused dgrad + LayerNorm-backward kernel (conv3x3_fwd_wres_kernel<., 4>) spends its cycles, per item.
Stamps serialise what the real kernel overlaps: read the SHARES, not the lengths (cdna_hip_programming.md, In-kernel stamps)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adunet_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
n, hw, c = 64, 256, 64
g = torch.Generator().manual_seed(1)
dz = (torch.rand((n, hw, hw, c), generator=g) * 2 - 1).to(device=dev, dtype=torch.bfloat16)
z = ((torch.rand((n, hw, hw, c), generator=g) * 2 - 1) * 1.5).to(device=dev, dtype=torch.bfloat16)
wk = ((torch.rand((3, 3, c, c), generator=g) * 2 - 1) * 0.05).to(dev)
gamma, beta = (torch.rand(c, generator=g) + 0.5).to(dev), (torch.rand(c, generator=g) - 0.5).to(dev)
mean = z.float().mean(-1).reshape(-1).contiguous()
rstd = torch.rsqrt(z.float().var(-1, unbiased=False) + 1e-3).reshape(-1).contiguous()
_, wd = ops.conv3x3_pack(wk, c, torch.bfloat16)
ws = ops.Workspace(dev, 64 << 20)
o = [torch.empty(c, device=dev) for _ in range(3)]
fn = lib.ad_dbg_set_stamp_buffer
fn.argtypes = [ctypes.c_void_p]
call = lambda: ops.conv3x3_dgrad_ln_bwd(dz, wd, z, mean, rstd, gamma, beta, o[0], o[1], o[2], ws)
for _ in range(200):
    call()
dbg = torch.zeros(512 * 10, dtype=torch.int64, device=dev)
fn(dbg.data_ptr())
call()
torch.cuda.synchronize()
fn(None)
d = dbg.view(512, 10).double().cpu()
d = d[d[:, 8] > 0]
tot, items = d[:, 8].mean(), d[:, 9].mean()
names = ["item set-up", "barrier waits", "MFMA phases (2 chunks)", "epilogue m-tile 0", "epilogue m-tile 1 (+ 2nd fetch issue)",
         "epilogue m-tile 2", "epilogue m-tile 3"]
print(f"{len(d)} workgroups, {items:.1f} items each, clock {tot / (d[:, 7].mean() * 10.0):.2f} GHz, {d[:, 7].mean() * 0.01:.1f} us in the kernel (stamped build)")
for i, nm in enumerate(names):
    print(f"   {nm:<40} {d[:, i].mean() / items:>8.0f} cycles per item ({d[:, i].mean() / tot:6.1%})")
print(f"   {'unstamped rest':<40} {(tot - d[:, :7].sum(1).mean()) / items:>8.0f}")
