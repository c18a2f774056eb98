#!/usr/bin/env python3
"""BASELINE config 5 under data parallelism, rehearsed with two ranks on ONE GPU over gloo (RCCL refuses two ranks per device):
`AdaptiveDepthBank.data_parallel()` gives every model of the bank its own bucketed gradient exchange; each rank trains on its half
of every batch of a mixed stream (SR 0.5 -> depth 3, segmentation, SR 0.3 -> depth 2, SR 0.5 again), eagerly and through the
per-model segmented graph replay.  The SR models (LayerNorm: no cross-sample statistic) must end with the weights of ONE process
training on the whole batches (fp32: summation order only); the segmentation model normalises with PER-REPLICA batch statistics,
as Keras' BatchNormalization does under a distribution strategy without sync-BN, so for it the contract is only
what holds for every model: the two ranks hold bitwise identical weights and eager == graph replay.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/dp2_bank_gloo_gpu_check.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from adunet_amd import multitask as M

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
P, per = 32, 2
rng = np.random.default_rng(8)

def sr():
    hr = rng.random((per * world, P, P, 3), dtype=np.float32)
    return np.clip(hr + 0.05 * rng.standard_normal(hr.shape, dtype=np.float32), 0, 1).astype(np.float32), hr

def seg():
    return rng.random((per * world, P, P, 3), dtype=np.float32), (rng.random((per * world, P, P, 1)) < 0.4).astype(np.float32)

stream = [("sr", 0.5, *sr()), ("seg", *seg()), ("sr", 0.3, *sr()), ("sr", 0.5, *sr()), ("seg", *seg())]
sl = slice(rank * per, (rank + 1) * per)

def shard(item):
    return item[:2] + tuple(a[sl] for a in item[2:]) if item[0] == "sr" else (item[0],) + tuple(a[sl] for a in item[1:])

def weights(bank):
    return {k: m.P.clone() for k, m in list(bank.sr.items()) + [("seg", bank.seg)]}

results = {}
for graphed in (False, True):
    bank = M.AdaptiveDepthBank(input_size=P, dtype=torch.float32, device=dev, learning_rate=1e-3, seg_depth=2).data_parallel(bucket_bytes=1 << 18)
    for item in stream:
        bank.train_on_batch(item[0], *shard(item)[1:], graphed=graphed)
    torch.cuda.synchronize()
    assert len(bank.dps) == 3 and all(dp.world == world for dp in bank.dps)
    results[graphed] = weights(bank)
    bank.close()
    del bank
ok = True
for graphed, w in results.items():
    for k, p in w.items():
        both = [torch.empty_like(p.cpu()) for _ in range(world)]
        dist.all_gather(both, p.cpu())
        ok &= torch.equal(both[0], both[1])
if rank == 0:
    single = M.AdaptiveDepthBank(input_size=P, dtype=torch.float32, device=dev, learning_rate=1e-3, seg_depth=2)
    for item in stream:
        single.train_on_batch(item[0], *item[1:], graphed=False)
    torch.cuda.synchronize()
    want = weights(single)
    for graphed, w in results.items():
        for k, p in w.items():
            err = float((p - want[k]).abs().max() / want[k].abs().max())
            print(f"{'graph' if graphed else 'eager'} {k}: max |P_dp - P_single| / max|P| = {err:.3e}"
                  + ("   (per-replica BatchNorm statistics: not expected to match)" if k == "seg" else ""), flush=True)
            ok &= k == "seg" or err < 1e-5
    same = all(torch.equal(results[False][k], results[True][k]) for k in want)
    print("eager == graph bitwise:", same, flush=True)
    ok &= same
t = torch.tensor([1.0 if ok else 0.0])
dist.all_reduce(t, op=dist.ReduceOp.MIN)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if t.item() == 1.0 else 1)
