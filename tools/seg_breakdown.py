#!/usr/bin/env python3
"""Per-op-family HIP-event table of one eager train step of the LayerNorm / Conv2DTranspose segmentation model
(build_unet depth 4, 64 base channels, 256 x 256, batch 16, bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adunet_amd import ops, seg_model as S
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "ln"
m = (S.build_unet(256, 1, 64, 4, dtype=torch.bfloat16, device=dev) if kind == "ln"
     else S.build_adaptive_depth_unet(256, 64, 4, dtype=torch.bfloat16, device=dev))
proto = S.PROTOCOLS["B"]
m.compile(optimizer=S.build_optimizer(proto, steps_per_epoch=100, epochs=2), loss=proto.loss_builder())
m._require_device()
b = 16
img = rng.random((b, 256, 256, 3), dtype=np.float32)
mask = (rng.random((b, 256, 256, 1)) < 0.35).astype(np.float32)
for _ in range(3):
    m.train_on_batch(img, mask)
torch.cuda.synchronize()
timer = ops.KernelTimer()
ops.set_timer(timer)
for _ in range(3):
    m.train_on_batch(img, mask)
torch.cuda.synchronize()
ops.set_timer(None)
summ = timer.summary()
tot = sum(v[1] for v in summ.values()) / 3
for k, (cnt, ms) in sorted(summ.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:<28}{cnt / 3:>8.1f}{ms / 3:>10.3f} ms{ms / 3 / tot:>8.1%}")
print(f"{'(sum)':<28}{'':>8}{tot:>10.3f} ms")
