#!/usr/bin/env python3
"""Fused conv+LayerNorm+ReLU launch against the two separate launches (64 -> 64 at 256x256, batch 64, bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adunet_amd import ops

dev = torch.device("cuda:0")
n, hw, c = 64, 256, 64
x = torch.randn((n, hw, hw, c), device=dev).bfloat16()
w = torch.randn((3, 3, c, c), device=dev) * 0.05
wf, _ = ops.conv3x3_pack(w, c, torch.bfloat16, want_dgrad=False)
b = torch.zeros(c, device=dev); g = torch.ones(c, device=dev); be = torch.zeros(c, device=dev)

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

t_conv = timeit(lambda: ops.conv3x3_fwd(x, None, wf, b, c))
z = ops.conv3x3_fwd(x, None, wf, b, c)
t_ln = timeit(lambda: ops.layernorm_relu_fwd(z, g, be))
t_fused = timeit(lambda: ops.conv3x3_ln_relu_fwd(x, None, wf, b, g, be, c))
print(f"conv {t_conv:.3f} ms + ln {t_ln:.3f} ms = {t_conv + t_ln:.3f} ms;  fused {t_fused:.3f} ms")
