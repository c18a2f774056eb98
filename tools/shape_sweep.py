#!/usr/bin/env python3
"""Train-step smoke + throughput over the reference's experiment shapes (scale, depth, batch), bf16, graph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adunet_amd.model import Adam, build_losses_and_metrics, build_super_resolution_unet

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for scale, depth, patch, batch in [(0.6, 4, 256, 32), (0.7, 5, 256, 8), (0.2, 1, 256, 64), (0.3, 2, 256, 64), (0.4, 3, 256, 32),
                                   (0.8, 5, 256, 8), (0.5, 3, 256, 64), (0.25, 4, 512, 16), (0.5, 2, 128, 4)]:
    model, info = build_super_resolution_unet(scale, depth_override=depth, input_size=patch, dtype=torch.bfloat16, device=dev)
    loss, metrics = build_losses_and_metrics("charbonnier")
    model.compile(optimizer=Adam(1e-4), loss=loss, metrics=metrics)
    model._require_device()
    model.set_weights(model.initial_weights(np.random.default_rng(1), head_uniform=0.05))
    hr = rng.random((batch, patch, patch, 3), dtype=np.float32)
    lr = np.clip(hr + 0.05 * rng.standard_normal(hr.shape, dtype=np.float32), 0, 1).astype(np.float32)
    l0 = float(model.train_on_batch(lr, hr)[0])
    step = model.make_graphed_train_step(lr, hr)
    x, y = torch.from_numpy(lr).to(dev), torch.from_numpy(hr).to(dev)
    for _ in range(2): step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): l = step(x, y)[0]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"scale {scale} depth {depth} P{patch} b{batch}: sizes {model.sizes if hasattr(model, 'sizes') else ''} "
          f"{batch / dt:8.1f} img/s  {dt * 1e3:7.2f} ms  loss {l0:.4f} -> {float(l):.4f}", flush=True)
    assert np.isfinite(float(l))
    del model, step
    torch.cuda.empty_cache()
