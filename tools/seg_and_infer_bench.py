#!/usr/bin/env python3
"""Throughput of the rows next to the headline: SR inference (forward only) and the two segmentation models' train
step (tier 2: BatchNorm / max-pool / bilinear x2 and LayerNorm / Conv2DTranspose), bf16, eager and graph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adunet_amd import seg_model as S
from adunet_amd.model import Adam, build_losses_and_metrics, build_super_resolution_unet

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)

def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

model, _ = build_super_resolution_unet(0.25, depth_override=4, input_size=256, dtype=torch.bfloat16, device=dev)
loss, metrics = build_losses_and_metrics("charbonnier")
model.compile(optimizer=Adam(1e-4), loss=loss, metrics=metrics)
model._require_device()
x = torch.from_numpy(rng.random((64, 256, 256, 3), dtype=np.float32)).to(dev)
dt = timeit(lambda: model(x, training=False))
print(f"SR inference scale 0.25 depth 4 P256 b64: {64 / dt:8.1f} img/s  {dt * 1e3:6.2f} ms")
del model

for name, build, b in (("adaptive_unet depth4 c64 (BN, maxpool, bilinear)", lambda: S.build_adaptive_depth_unet(256, 64, 4, dtype=torch.bfloat16, device=dev), 16),
                       ("unet depth4 c64 (LN, Conv2DTranspose)", lambda: S.build_unet(256, 1, 64, 4, dtype=torch.bfloat16, device=dev), 16)):
    m = build()
    proto = S.PROTOCOLS["B"]
    m.compile(optimizer=S.build_optimizer(proto, steps_per_epoch=100, epochs=2), loss=proto.loss_builder())
    m._require_device()
    img = rng.random((b, 256, 256, 3), dtype=np.float32)
    mask = (rng.random((b, 256, 256, 1)) < 0.35).astype(np.float32)
    dt = timeit(lambda: m.train_on_batch(img, mask))
    step = m.make_graphed_train_step(img, mask)
    xi, xm = m._graph_inputs(img, mask)
    dg = timeit(lambda: step(xi, xm))
    print(f"seg train {name} P256 b{b}: eager {b / dt:8.1f} img/s {dt * 1e3:6.2f} ms | graph replay {b / dg:8.1f} img/s {dg * 1e3:6.2f} ms")
    del m, step
