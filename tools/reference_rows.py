#!/usr/bin/env python3
"""The reference's own published training rows, run like for like.

BASELINE.md 1.1 / SURVEY 6: the only throughput the reference publishes are the Keras ms/step values of its Experiment 2
(`Super_resolution/experiments/experiment_2_adaptive_depth/csv_logs/*/epoch_metrics.csv`; scale -> depth and batch from
`run_experiment_adaptive_depth.sh:36-66`), 256 x 256 patches, `mixed_float16`, one GPU its scripts size for "a 2080 Ti".  This
runs the SAME rows -- same scale, depth, batch, patch and precision policy (fp16 kernels + Keras dynamic loss scaling), one
graph-replayed train step on a resident synthetic batch -- and, beside them, the same model at a batch that fills an MI355X.
Not the headline metric (that is K2' at batch 64 in bf16, bench.py) and not a `vs_baseline`: other hardware, other batch sizes.

    python tools/reference_rows.py [--dtype f16|bf16]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from adunet_amd.model import Adam, build_losses_and_metrics, build_super_resolution_unet  # noqa: E402

# (scale, depth, batch, reference ms/step first..last epoch) -- BASELINE.md 1.1
ROWS = [(0.2, 1, 8, (380, 381)), (0.3, 2, 8, (509, 511)), (0.4, 3, 6, (498, 498)), (0.5, 3, 4, (428, 445)),
        (0.6, 4, 3, (557, 577)), (0.7, 5, 2, (877, 895)), (0.8, 5, 1, (976, 977))]
FILL = {1: 64, 2: 64, 3: 64, 4: 32, 5: 8}          # a batch that gives every CU work (tools/scale_sweep.py)


def rate(scale, depth, batch, dtype, dev, steps=10):
    model, _ = build_super_resolution_unet(scale, depth_override=depth, input_size=256, dtype=dtype, device=dev)
    loss, metrics = build_losses_and_metrics("charbonnier")
    model.compile(optimizer=Adam(1e-4), loss=loss, metrics=metrics)
    model._require_device()
    model.set_weights(model.initial_weights(np.random.default_rng(1234), head_uniform=0.05))
    lr, hr = bench.synth_batch(0, batch, 256, dev)
    step = model.make_graphed_train_step(lr, hr)
    for _ in range(3):
        step(lr, hr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step(lr, hr)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    fwd, first = bench.conv_flops_per_image(model)
    frac = batch / ms * 1e3 * (3 * fwd - first) / 1e12 / bench.PEAK_BF16_TFLOPS
    loss_v = float(out[0])
    del model, step
    torch.cuda.empty_cache()
    return ms, batch / ms * 1e3, frac, loss_v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    args = ap.parse_args()
    dtype = {"f16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    dev = torch.device("cuda:0")
    print(f"policy: {args.dtype}{' + dynamic loss scaling (mixed_float16)' if args.dtype == 'f16' else ''}; patch 256; graph replay; resident batch")
    print(f"{'scale':>5}{'depth':>6}{'batch':>6} |{'reference ms/step':>18}{'img/s':>8} |{'here ms/step':>13}{'img/s':>9}{'x':>7}{'of peak':>9} |"
          f"{'batch':>6}{'ms/step':>9}{'img/s':>9}{'of peak':>9}")
    for scale, depth, batch, (r0, r1) in ROWS:
        ref_ms = 0.5 * (r0 + r1)
        ref_ips = batch / ref_ms * 1e3
        ms, ips, frac, lv = rate(scale, depth, batch, dtype, dev)
        assert np.isfinite(lv)
        fb = FILL[depth]
        ms2, ips2, frac2, _ = rate(scale, depth, fb, dtype, dev)
        print(f"{scale:>5.1f}{depth:>6}{batch:>6} |{f'{r0}-{r1}':>18}{ref_ips:>8.1f} |{ms:>13.2f}{ips:>9.0f}{ips / ref_ips:>7.0f}{frac:>9.3f} |"
              f"{fb:>6}{ms2:>9.2f}{ips2:>9.0f}{frac2:>9.3f}", flush=True)


if __name__ == "__main__":
    main()
