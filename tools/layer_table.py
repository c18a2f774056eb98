"""Per-launch table of the conv kernels inside one eager train step of a workload: shape, time, TFLOP/s, fraction of peak.

    python tools/layer_table.py --workload E2s06 [--batch B]        (or --workload scale,depth,patch,batch)

The ops wrappers are patched to bracket each call with HIP events (launch stream); two eager steps are timed, the second
is printed.  Shows which launches of a pyramid sit far below the roof (odd map widths, small maps, wide channels)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from adunet_amd import ops  # noqa: E402
from bench import PEAK_BF16_TFLOPS, WORKLOADS, synth_batch  # noqa: E402

rows = []


def wrap(name, shape_of):
    fn = getattr(ops, name)

    def inner(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kw)
        e1.record()
        rows.append((name, shape_of(*args, **kw), e0, e1))
        return out

    setattr(ops, name, inner)


def conv_shape(x1, x2, *rest, **kw):
    n, h, w, c1 = x1.shape
    c2 = x2.shape[-1] if x2 is not None else 0
    return n, h, w, c1 + c2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="E2s06")
    ap.add_argument("--batch", type=int, default=None)
    args = ap.parse_args()
    from adunet_amd.model import Adam, build_losses_and_metrics, build_super_resolution_unet
    if args.workload in WORKLOADS:
        scale, depth, patch, batch = WORKLOADS[args.workload]
    else:                                            # "scale,depth,patch,batch", e.g. 0.4,3,256,64
        f = args.workload.split(",")
        scale, depth, patch, batch = float(f[0]), int(f[1]), int(f[2]), int(f[3])
    batch = args.batch or batch
    dev = torch.device("cuda:0")
    model, _ = build_super_resolution_unet(scale, depth_override=depth, input_size=patch, dtype=torch.bfloat16, device=dev)
    loss, metrics = build_losses_and_metrics("charbonnier")
    model.compile(optimizer=Adam(1e-4), loss=loss, metrics=metrics)
    model._require_device()
    model.set_weights(model.initial_weights(np.random.default_rng(1234), head_uniform=0.05))
    wrap("conv3x3_fwd", lambda x1, x2, w, b, cout, **kw: conv_shape(x1, x2) + (cout,))
    wrap("conv3x3_ln_relu_fwd", lambda x1, x2, w, b, g, be, cout, **kw: conv_shape(x1, x2) + (cout,))
    wrap("conv3x3_wgrad", lambda x1, x2, dz, dw, cin, ws: conv_shape(x1, x2) + (dz.shape[-1],))
    wrap("conv3x3_dgrad_relu", lambda dz, wd, u, db, cout, ws: tuple(dz.shape) + (cout,))
    wrap("conv3x3_dgrad_ln_bwd", lambda dz, wd, z, *r: tuple(dz.shape) + (z.shape[-1],))
    wrap("pw_gemm", lambda x, bank, n_out: (x.shape[0], x.shape[1], x.shape[2], x.shape[3], n_out))
    lr, hr = synth_batch(0, batch, patch, dev)
    for _ in range(2):
        rows.clear()
        model.train_on_batch(lr, hr)
        torch.cuda.synchronize()
    print(f"{'op':<24}{'n x h x w':>16}{'cin':>6}{'cout':>6}{'ms':>9}{'TFLOP/s':>9}{'frac':>7}")
    tot = {}
    for name, (n, h, w, cin, cout), e0, e1 in rows:
        ms = e0.elapsed_time(e1)
        taps = 1 if name == "pw_gemm" else 9
        fl = 2.0 * n * h * w * taps * cin * cout
        print(f"{name:<24}{f'{n}x{h}x{w}':>16}{cin:>6}{cout:>6}{ms:>9.3f}{fl / ms / 1e9:>9.0f}{fl / ms / 1e9 / PEAK_BF16_TFLOPS:>7.3f}")
        t = tot.setdefault((name, h), [0.0, 0.0])
        t[0] += ms
        t[1] += fl
    print("\nby op and map width:")
    for (name, h), (ms, fl) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        print(f"{name:<24}{h:>6}{ms:>9.3f} ms{fl / ms / 1e9 / PEAK_BF16_TFLOPS:>7.3f}")


if __name__ == "__main__":
    main()
