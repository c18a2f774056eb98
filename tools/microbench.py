#!/usr/bin/env python3
"""Per-shape timing of the conv kernels (HIP events), e.g. the BASELINE micro-kernel target:
3x3 conv 64->64 at 256x256, batch 32, bf16: fwd / dgrad / wgrad TFLOP/s and fraction of the 2.5 PF peak."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adunet_amd import ops  # noqa: E402

SHAPES = {
    "mu": [(32, 256, 64, 0, 64)],
    "k2p": [(64, 256, 32, 0, 64), (64, 256, 64, 0, 64), (64, 256, 128, 0, 64), (64, 256, 64, 64, 64),
            (64, 64, 64, 0, 128), (64, 64, 128, 0, 128), (64, 64, 256, 0, 128), (64, 64, 128, 128, 128),
            (64, 16, 128, 0, 256), (64, 16, 256, 0, 256), (64, 16, 512, 0, 256),
            (64, 4, 256, 0, 512), (64, 4, 512, 0, 512), (64, 4, 1024, 0, 512), (64, 1, 512, 0, 1024), (64, 1, 1024, 0, 1024)],
    # the LayerNorm / Conv2DTranspose segmentation model at batch 16 (conv blocks, decoder concat convs, transposed convs as
    # 1x1 GEMMs run through the same kernels)
    "seg": [(16, 256, 64, 0, 64), (16, 128, 64, 0, 128), (16, 128, 128, 0, 128), (16, 64, 128, 0, 256), (16, 64, 256, 0, 256),
            (16, 32, 256, 0, 512), (16, 32, 512, 0, 512), (16, 16, 512, 0, 1024), (16, 16, 1024, 0, 1024),
            (16, 16, 1024, 0, 2048), (16, 32, 512, 0, 1024), (16, 64, 256, 0, 512), (16, 128, 128, 0, 256),
            (16, 32, 512, 512, 512), (16, 64, 256, 256, 256), (16, 128, 128, 128, 128), (16, 256, 64, 64, 64)],
    # 4x4 maps at larger batches (where does conv3x3_map4_kernel stop paying?)
    "small_n": [(256, 4, 512, 0, 512), (1024, 4, 512, 0, 512), (4096, 4, 512, 0, 512), (1024, 4, 1024, 0, 512), (4096, 4, 256, 0, 512)],
    # the 4x4 and 1x1 levels of K2' 
    "small": [(64, 4, 256, 0, 512), (64, 4, 512, 0, 512), (64, 4, 1024, 0, 512), (64, 4, 512, 512, 512),
              (64, 1, 512, 0, 1024), (64, 1, 1024, 0, 1024)],
}


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="mu", choices=sorted(SHAPES))
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    ws = ops.Workspace(dev, 256 << 20)
    print(f"{'n':>3} {'hw':>4} {'c1':>5} {'c2':>5} {'cout':>5} | {'fwd ms':>8} {'TF/s':>7} | {'dgrad ms':>8} {'TF/s':>7} | {'wgrad ms':>8} {'TF/s':>7}")
    for n, hw, c1, c2, cout in SHAPES[args.set]:
        cin = c1 + c2
        x1 = torch.randn((n, hw, hw, c1), device=dev).to(dtype)
        x2 = torch.randn((n, hw, hw, c2), device=dev).to(dtype) if c2 else None
        w = torch.randn((3, 3, cin, cout), device=dev) * 0.05
        b = torch.zeros(cout, device=dev)
        wf, wd = ops.conv3x3_pack(w, cin, dtype)
        dz = torch.randn((n, hw, hw, cout), device=dev).to(dtype)
        dw = torch.empty((3, 3, cin, cout), device=dev)
        flops = 2.0 * n * hw * hw * 9 * cin * cout
        t_f = timeit(lambda: ops.conv3x3_fwd(x1, x2, wf, b, cout), args.iters)
        t_d = (timeit(lambda: ops.conv3x3_fwd(dz, None, wd, None, cin, split=c1 if c2 else None), args.iters)
               if cin % 64 == 0 else float("inf"))
        t_w = timeit(lambda: ops.conv3x3_wgrad(x1, x2, dz, dw, cin, ws), args.iters)
        print(f"{n:>3} {hw:>4} {c1:>5} {c2:>5} {cout:>5} | {t_f:>8.3f} {flops / t_f / 1e9:>7.0f} | {t_d:>8.3f} {flops / t_d / 1e9:>7.0f} | "
              f"{t_w:>8.3f} {flops / t_w / 1e9:>7.0f}", flush=True)


if __name__ == "__main__":
    main()
