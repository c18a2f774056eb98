#!/usr/bin/env python3
"""Diagnostic (build with AD_CFLAGS=-DAD_STAMP): where MFMA wave 0 of every workgroup of the wave-specialised forward
kernels spends its cycles.  usage: stamps_ws.py [n hw cin cout]   (default: 64 x 256 x 256, 64 -> 64)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adunet_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
n, hw, cin, cout = 64, 256, 64, 64
if len(sys.argv) > 1:
    n, hw, cin, cout = map(int, sys.argv[1:5])
x = torch.randn((n, hw, hw, cin), device=dev).bfloat16()
w = torch.randn((3, 3, cin, cout), device=dev) * 0.05
wf, _ = ops.conv3x3_pack(w, cin, torch.bfloat16, want_dgrad=False)
b = torch.zeros(cout, device=dev)
gamma, beta = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
fn = lib.ad_dbg_set_stamp_buffer
fn.argtypes = [ctypes.c_void_p]
names = ["item set-up", "barrier waits", "MFMA phases", "pack", "final drain", "items", "-", "wall (10 ns)", "TOTAL cycles"]


def run(label, call):
    dbg = torch.zeros(512 * 9, dtype=torch.int64, device=dev)
    for _ in range(20):
        call()
    fn(dbg.data_ptr())
    call()
    torch.cuda.synchronize()
    fn(None)
    d = dbg.view(512, 9).double().cpu()
    d = d[d[:, 8] > 0]
    tot = d[:, 8].mean()
    print(f"== {label}: {len(d)} workgroups, {d[:, 5].mean():.1f} items each, clock {tot / (d[:, 7].mean() * 10.0):.2f} GHz, "
          f"{d[:, 7].mean() * 0.01:.1f} us in the kernel")
    for i in (0, 1, 2, 3, 4):
        print(f"   {names[i]:<14} {d[:, i].mean():>10.0f} cycles ({d[:, i].mean() / tot:6.1%})  per item {d[:, i].mean() / d[:, 5].mean():>7.0f}")


run("bias epilogue", lambda: ops.conv3x3_fwd(x, None, wf, b, cout))
if cout == 64:
    run("fused LayerNorm + ReLU", lambda: ops.conv3x3_ln_relu_fwd(x, None, wf, b, gamma, beta, cout, 1e-3))
