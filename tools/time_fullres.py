#!/usr/bin/env python3
"""Times the full-resolution 64 -> 64 conv launches of K2' (plain and fused LayerNorm) with HIP events, after a warm-up
that lets the clock settle.  Used with ADUNET_LIB=ab/<variant>.so for diagnostic builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adunet_amd import ops
dev = torch.device("cuda:0")
n, hw, cin, cout = 64, 256, 64, 64
x = torch.randn((n, hw, hw, cin), device=dev).bfloat16()
w = torch.randn((3, 3, cin, cout), device=dev) * 0.05
wf, wd = ops.conv3x3_pack(w, cin, torch.bfloat16)
b = torch.zeros(cout, device=dev)
gamma, beta = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)


def timeit(fn, iters=100):
    for _ in range(150):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print(os.environ.get("ADUNET_LIB", "in-tree"))
print(f"  plain 64->64      {timeit(lambda: ops.conv3x3_fwd(x, None, wf, b, cout)):8.1f} us")
print(f"  fused LN 64->64   {timeit(lambda: ops.conv3x3_ln_relu_fwd(x, None, wf, b, gamma, beta, cout, 1e-3)):8.1f} us")
