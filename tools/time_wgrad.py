#!/usr/bin/env python3
"""HIP-event time of the full-resolution weight-gradient launch (64 x 256 x 256, 64 -> 64, bf16: conv3x3_wgrad_ws_kernel + its slab
reduce) after a clock-settling warm-up.  With ADUNET_LIB=ab/<variant>.so for diagnostic builds (e.g. -DAD_NO_SLAB: the same
launch without the 37.7 MB slab write).

r04 result (the -DAD_NO_SLAB variant was a local patch that replaced the slab stores at the end of conv3x3_wgrad_ws_kernel by a
register keep-alive; not kept in the tree): 267 us with the slab write, 264 us without -- the 147 KB per workgroup written as
4-byte stores cost 1 % of the launch, so neither 16-byte slab stores nor fewer slabs are worth building."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adunet_amd import ops
dev = torch.device("cuda:0")
n, hw, c = 64, 256, 64
x = torch.randn((n, hw, hw, c), device=dev).bfloat16()
dz = torch.randn((n, hw, hw, c), device=dev).bfloat16()
dw = torch.empty((3, 3, c, c), device=dev)
ws = ops.Workspace(dev, 256 << 20)
fn = lambda: ops.conv3x3_wgrad(x, None, dz, dw, c, ws)
for _ in range(200):
    fn()
torch.cuda.synchronize()
best = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fn()
    e1.record()
    torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 100 * 1e3)
print(os.environ.get("ADUNET_LIB", "in-tree"), " ".join(f"{b:.1f}" for b in best), "us per launch (kernel + reduce)")
