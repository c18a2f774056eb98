#!/usr/bin/env python3
"""Are the full-resolution launches memory-bound?  The same launch on batches whose tensors FIT the 256 MiB Infinity Cache (all
operands re-read / re-written on-die from the second iteration on) against the benchmark batch (537 MB per tensor: every byte
crosses HBM): microseconds per IMAGE.  64 x 256 x 256 x 64 -> 64, bf16, back-to-back launches after a warm-up."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adunet_amd import ops
dev = torch.device("cuda:0")
hw, c = 256, 64
g = torch.Generator().manual_seed(1)
wk = ((torch.rand((3, 3, c, c), generator=g) * 2 - 1) * 0.05).to(dev)
wf, wd = ops.conv3x3_pack(wk, c, torch.bfloat16)
gamma, beta, b = (torch.rand(c, generator=g) + 0.5).to(dev), (torch.rand(c, generator=g) - 0.5).to(dev), torch.zeros(c, device=dev)
ws = ops.Workspace(dev, 64 << 20)
o = [torch.empty(c, device=dev) for _ in range(3)]
dw = torch.empty_like(wk)


def timeit(fn, iters):
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print(f"{'images':>7}{'tensor MB':>11} | us per image: {'plain conv':>11}{'conv+LN fwd':>13}{'dgrad+LN bwd':>14}{'wgrad':>9}")
for n in (4, 8, 16, 32, 64):
    x = (torch.rand((n, hw, hw, c), generator=g) * 2 - 1).to(device=dev, dtype=torch.bfloat16)
    dz = (torch.rand((n, hw, hw, c), generator=g) * 2 - 1).to(device=dev, dtype=torch.bfloat16)
    mean = x.float().mean(-1).reshape(-1).contiguous()
    rstd = torch.rsqrt(x.float().var(-1, unbiased=False) + 1e-3).reshape(-1).contiguous()
    it = max(40, 1600 // n)
    t0 = timeit(lambda: ops.conv3x3_fwd(x, None, wf, b, c), it)
    t1 = timeit(lambda: ops.conv3x3_ln_relu_fwd(x, None, wf, b, gamma, beta, c), it)
    t2 = timeit(lambda: ops.conv3x3_dgrad_ln_bwd(dz, wd, x, mean, rstd, gamma, beta, o[0], o[1], o[2], ws), it)
    t3 = timeit(lambda: ops.conv3x3_wgrad(x, None, dz, dw, c, ws), it)
    print(f"{n:>7}{n * hw * hw * c * 2 / 1e6:>11.0f} |               {t0 / n:>11.2f}{t1 / n:>13.2f}{t2 / n:>14.2f}{t3 / n:>9.2f}")
