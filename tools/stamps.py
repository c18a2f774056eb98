#!/usr/bin/env python3
"""Diagnostic (build with AD_CFLAGS=-DAD_STAMP): where one workgroup of conv3x3_fwd_kernel spends its cycles.
Only the generic kernel carries the stamps: use a shape the wave-specialised kernels do not take (default 2 x 64 x 64)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adunet_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
n, hw, cin, cout = 2, 64, 64, 64
if len(sys.argv) > 1:
    n, hw, cin, cout = map(int, sys.argv[1:5])
x = torch.randn((n, hw, hw, cin), device=dev).bfloat16()
w = torch.randn((3, 3, cin, cout), device=dev) * 0.05
wf, _ = ops.conv3x3_pack(w, cin, torch.bfloat16, want_dgrad=False)
b = torch.zeros(cout, device=dev)
dbg = torch.zeros(512 * 9, dtype=torch.int64, device=dev)
fn = lib.ad_dbg_set_stamp_buffer
fn.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.conv3x3_fwd(x, None, wf, b, cout)
fn(dbg.data_ptr())
ops.conv3x3_fwd(x, None, wf, b, cout)
torch.cuda.synchronize()
fn(None)
d = dbg.view(512, 9).double().cpu()
d = d[d[:, 8] > 0]
names = ["prologue", "vmwait+LDS store", "gtab build", "barrier A", "issue prefetch", "MFMA phase", "barrier B", "epilogue", "TOTAL"]
tot = d[:, 8].mean()
for i, nm in enumerate(names):
    print(f"{nm:<18} mean {d[:, i].mean():>10.0f}  ({d[:, i].mean() / tot:6.1%})   min {d[:, i].min():>9.0f} max {d[:, i].max():>9.0f}")
