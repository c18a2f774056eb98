#!/usr/bin/env python3
"""HIP-event timing of the dgrad of a full-resolution 64 -> 64 conv followed by the LayerNorm / ReLU backward of the layer
below it: fused launch (ad_conv3x3_dgrad_ln_bwd) against the two launches it replaces (K2' shape: 64 x 256 x 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adunet_amd import ops
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


def timeit(fn, iters=60):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def two():
    d = ops.conv3x3_fwd(dz, None, wd, None, c)
    return ops.layernorm_relu_bwd(d, z, mean, rstd, gamma, beta, o[0], o[1], o[2], ws)


print(f"two launches  {timeit(two):8.1f} us")
print(f"  dgrad alone {timeit(lambda: ops.conv3x3_fwd(dz, None, wd, None, c)):8.1f} us")
if ops.conv3x3_dgrad_ln_bwd_is_fused(dz, c):
    print(f"fused         {timeit(lambda: ops.conv3x3_dgrad_ln_bwd(dz, wd, z, mean, rstd, gamma, beta, o[0], o[1], o[2], ws)):8.1f} us")
