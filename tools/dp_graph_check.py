#!/usr/bin/env python3
"""World-size-1 RCCL check of the segmented-graph data-parallel step (diagnostic)."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29741")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from adunet_amd.model import Adam, build_losses_and_metrics, build_super_resolution_unet
from adunet_amd.parallel import DataParallel
rng = np.random.default_rng(22)
def synth(n, p):
    hr = rng.random((n, p, p, 3), dtype=np.float32)
    return np.clip(hr + 0.05 * rng.standard_normal(hr.shape, dtype=np.float32), 0, 1).astype(np.float32), hr
b0 = synth(3, 32)
try:
    model, _ = build_super_resolution_unet(0.5, depth_override=2, input_size=32, dtype=torch.bfloat16, device=dev)
    loss, metrics = build_losses_and_metrics("charbonnier")
    model.compile(optimizer=Adam(1e-3), loss=loss, metrics=metrics)
    model._require_device()
    dp = DataParallel(model, bucket_bytes=1 << 20)
    print("buckets", len(dp.buckets), flush=True)
    step = model.make_graphed_train_step(*b0)
    print("segments", len(step.segments), flush=True)
    for i in range(3):
        print(float(step(*b0)[0]), flush=True)
except Exception:
    traceback.print_exc(); sys.stdout.flush(); sys.stderr.flush()
    os._exit(1)
torch.cuda.synchronize()
print("ok", flush=True)
os._exit(0)
