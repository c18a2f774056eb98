#!/usr/bin/env python3
"""Probe: does an MFMA-bound wgrad launch overlap with an HBM-bound LayerNorm-backward launch on a second stream?
(full-resolution 64-channel layer of K2' at batch 64; sum of the two alone vs both issued concurrently)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adunet_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n, hw, c = 64, 256, 64
    g = torch.Generator().manual_seed(1)
    x = (torch.rand((n, hw, hw, c), generator=g) * 2 - 1).to(dev, torch.bfloat16)
    dz = (torch.rand((n, hw, hw, c), generator=g) * 2 - 1).to(dev, torch.bfloat16)
    z = (torch.rand((n, hw, hw, c), generator=g) * 2 - 1).to(dev, torch.bfloat16)
    da = (torch.rand((n, hw, hw, c), generator=g) * 2 - 1).to(dev, torch.bfloat16)
    mean = torch.zeros(n * hw * hw, device=dev)
    rstd = torch.ones(n * hw * hw, device=dev)
    gam, bet = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    dgam, dbet, dbias = (torch.empty(c, device=dev) for _ in range(3))
    w = ((torch.rand((3, 3, c, c), generator=g) * 2 - 1) * 0.05).to(dev)
    wf, wd = ops.conv3x3_pack(w, c, torch.bfloat16)
    dw = torch.empty_like(w)
    ws1, ws2 = ops.Workspace(dev, 128 << 20), ops.Workspace(dev, 16 << 20)
    side = torch.cuda.Stream()
    wgrad = lambda: ops.conv3x3_wgrad(x, None, dz, dw, c, ws1)
    dgrad = lambda: ops.conv3x3_fwd(dz, None, wd, None, c)
    lnb = lambda: ops.layernorm_relu_bwd(da, z, mean, rstd, gam, bet, dgam, dbet, dbias, ws2)

    def timed(fn, iters=30):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    def both(a, b):
        def run():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                a()
            b()
            torch.cuda.current_stream().wait_stream(side)
        return run

    t_w, t_d, t_l = timed(wgrad), timed(dgrad), timed(lnb)
    print(f"alone: wgrad {t_w:.3f} ms, dgrad {t_d:.3f} ms, ln_bwd {t_l:.3f} ms")
    print(f"wgrad || ln_bwd: {timed(both(wgrad, lnb)):.3f} ms (sum {t_w + t_l:.3f})")
    print(f"dgrad || ln_bwd: {timed(both(dgrad, lnb)):.3f} ms (sum {t_d + t_l:.3f})")
    print(f"wgrad || dgrad : {timed(both(wgrad, dgrad)):.3f} ms (sum {t_w + t_d:.3f})")


if __name__ == "__main__":
    main()
