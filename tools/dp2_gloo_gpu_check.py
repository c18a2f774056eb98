#!/usr/bin/env python3
"""Two data-parallel ranks sharing ONE GPU over gloo (RCCL refuses two ranks on one device): exercises the world-size-2
logic of DataParallel + segmented graph replay on real device tensors.  Each rank trains on its half of a batch; the
result must equal one process training on the whole batch (LayerNorm has no cross-sample statistic).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 \
        tools/dp2_gloo_gpu_check.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from adunet_amd.model import Adam, build_losses_and_metrics, build_super_resolution_unet
from adunet_amd.parallel import DataParallel

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
rng = np.random.default_rng(5)
steps = 3
per = 2
batches = []
for _ in range(steps):
    hr = rng.random((per * world, 32, 32, 3), dtype=np.float32)
    lr = np.clip(hr + 0.05 * rng.standard_normal(hr.shape, dtype=np.float32), 0, 1).astype(np.float32)
    batches.append((lr, hr))

def make():
    m, _ = build_super_resolution_unet(0.5, depth_override=2, input_size=32, dtype=torch.float32, device=dev)
    loss, metrics = build_losses_and_metrics("charbonnier")
    m.compile(optimizer=Adam(1e-3), loss=loss, metrics=metrics)
    m._require_device()
    m.set_weights(m.initial_weights(np.random.default_rng(1), head_uniform=0.05))
    return m

results = {}
for mode in ("eager", "graph"):
    m = make()
    DataParallel(m, bucket_bytes=1 << 18)
    sl = slice(rank * per, (rank + 1) * per)
    if mode == "graph":
        step = m.make_graphed_train_step(batches[0][0][sl], batches[0][1][sl], capture_only=True)
    else:
        step = m.train_on_batch
    for lr, hr in batches:
        step(lr[sl], hr[sl])
    torch.cuda.synchronize()
    results[mode] = m.P.clone()
    del m, step
ref = None
if rank == 0:
    m = make()                               # one process, whole batch
    for lr, hr in batches:
        m.train_on_batch(lr, hr)
    ref = m.P.clone()
ok = True
if rank == 0:
    for mode, p in results.items():
        err = float((p - ref).abs().max() / ref.abs().max())
        print(f"{mode}: max |P_dp - P_single| / max|P| = {err:.3e}", flush=True)
        ok &= err < 1e-5
    print("eager == graph bitwise:", bool(torch.equal(results["eager"], results["graph"])), flush=True)
    ok &= bool(torch.equal(results["eager"], results["graph"]))
t = torch.tensor([1.0 if ok else 0.0])
dist.broadcast(t, src=0)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if t.item() == 1.0 else 1)
