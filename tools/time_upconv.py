"""Times the kernels of the factored up-conv (csrc/upconv.hip) one by one on the decoder levels of a workload.

    python tools/time_upconv.py [--workload K2p] [--iters 50]

HIP events on the launch stream around `iters` back-to-back launches (after a warm-up), per level: the 1x1 bank GEMM,
the forward gather, the backward gather, the dx GEMM and the bank weight gradient, with the bytes each must move."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adunet_amd import ops  # noqa: E402
from bench import WORKLOADS  # noqa: E402


def timed(fn, iters):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="K2p")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    from adunet_amd.model import build_super_resolution_unet
    scale, depth, patch, batch = WORKLOADS[args.workload]
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    dev = torch.device("cuda:0")
    model, _ = build_super_resolution_unet(scale, depth_override=depth, input_size=patch, dtype=dtype, device=dev)
    model._require_device()
    ws = ops.Workspace(dev)
    es = 2 if dtype != torch.float32 else 4
    print(f"{'level':<28}{'op':<20}{'ms':>9}{'GB':>8}{'TB/s':>8}{'TFLOP/s':>9}")
    for step in model._plan:
        if step[0] != "upconv":
            continue
        cs, lvl = step[1], step[2]
        src, dst = model.sizes[lvl + 1], cs.hw
        if cs.name not in model._banks:
            continue
        tab = model._upconv_tables(src, dst)
        x = (torch.rand((batch, src, src, cs.cin), device=dev) - 0.5).to(dtype)
        g = (torch.rand((batch, dst, dst, cs.cout), device=dev) - 0.5).to(dtype)
        bias = torch.zeros(cs.cout, device=dev)
        bf, bd = model._banks[cs.name]
        dw = torch.empty((3, 3, cs.cin, cs.cout), device=dev)
        yb = ops.pw_gemm(x, bf, 9 * cs.cout)
        dyb = ops.upconv_gather_bwd(g, tab)
        m = batch * src * src
        name = f"{src}->{dst} {cs.cin}->{cs.cout}"
        rows = [
            ("pw_gemm (Y)", lambda: ops.pw_gemm(x, bf, 9 * cs.cout), m * (cs.cin + 9 * cs.cout) * es, 2.0 * m * cs.cin * 9 * cs.cout),
            ("gather_fwd", lambda: ops.upconv_gather_fwd(yb, bias, tab), (yb.numel() + g.numel()) * es, 0.0),
            ("gather_bwd", lambda: ops.upconv_gather_bwd(g, tab), (yb.numel() + g.numel()) * es, 0.0),
            ("pw_gemm (dx)", lambda: ops.pw_gemm(dyb, bd, cs.cin), m * (cs.cin + 9 * cs.cout) * es, 2.0 * m * cs.cin * 9 * cs.cout),
            ("pw_wgrad", lambda: ops.upconv_bank_wgrad(x, dyb, dw, ws), m * (cs.cin + 9 * cs.cout) * es, 2.0 * m * cs.cin * 9 * cs.cout),
        ]
        for op, fn, nbytes, flops in rows:
            ms = timed(fn, args.iters)
            print(f"{name:<28}{op:<20}{ms:>9.4f}{nbytes / 1e9:>8.3f}{nbytes / ms / 1e9:>8.2f}{flops / ms / 1e9:>9.0f}")


if __name__ == "__main__":
    main()
