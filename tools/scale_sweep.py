#!/usr/bin/env python3
"""Train-step rate over the scales of the reference's Experiment 1 (depth from custom_depth_from_scale, capped so the bottleneck
stays at 2 048 channels), graph replay, bf16: images/s and the whole-step fraction of the MFMA peak, with and without the image
mosaic (ADUNET_NO_MOSAIC is read at library load, so each setting runs in its own process: tools/scale_sweep.py [--no-mosaic])."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--no-mosaic", action="store_true")
ap.add_argument("--scales", default="0.3,0.4,0.5,0.6,0.7,0.8")
args = ap.parse_args()
if args.no_mosaic:
    os.environ["ADUNET_NO_MOSAIC"] = "1"
import numpy as np, torch
import bench
from adunet_amd.custom_layers import custom_depth_from_scale
from adunet_amd.model import Adam, build_losses_and_metrics, build_super_resolution_unet
dev = torch.device("cuda:0")
for sc in [float(x) for x in args.scales.split(",")]:
    depth = min(custom_depth_from_scale(sc), 5)
    batch = {1: 64, 2: 64, 3: 64, 4: 32, 5: 8}[depth]
    model, _ = build_super_resolution_unet(sc, depth_override=depth, input_size=256, dtype=torch.bfloat16, device=dev)
    loss, metrics = build_losses_and_metrics("charbonnier")
    model.compile(optimizer=Adam(1e-4), loss=loss, metrics=metrics)
    lr, hr = bench.synth_batch(0, batch, 256, dev)
    step = model.make_graphed_train_step(lr, hr)
    for _ in range(3):
        step(lr, hr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step(lr, hr)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    fwd, first = bench.conv_flops_per_image(model)
    f_step = 3 * fwd - first
    sizes = sorted({cs.hw for cs in model.convs.values()}, reverse=True)
    print(f"scale {sc:.1f} depth {depth} batch {batch:3d}  maps {sizes}  {batch / ms * 1e3:8.0f} img/s  {ms:7.2f} ms  "
          f"frac_step {batch / ms * 1e3 * f_step / 1e12 / bench.PEAK_BF16_TFLOPS:.3f}", flush=True)
    del model, step
    torch.cuda.empty_cache()
