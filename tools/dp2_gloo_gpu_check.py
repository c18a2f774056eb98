#!/usr/bin/env python3
"""Two data-parallel ranks sharing ONE GPU over gloo (RCCL refuses two ranks on one device): exercises the world-size-2
logic of DataParallel + segmented graph replay on real device tensors.  Each rank trains on its half of a batch; the
result must equal one process training on the whole batch (LayerNorm has no cross-sample statistic).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 \
        tools/dp2_gloo_gpu_check.py [--dtype f32|bf16|f16]

--dtype f16 is the reference's policy (mixed_float16 + Keras dynamic loss scaling, train_adaptive_unet.py:471-477) and adds
the case the exchange order exists for: in step 1 ONE rank's half batch overflows (an input of 1e30 becomes inf in half
precision, its gradients NaN).  The finiteness check runs AFTER the all-reduce (model._apply_gradients), so both ranks must
see the non-finite sum, both skip the step, halve the scale, and end with identical scaler state and identical weights.
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from adunet_amd.model import Adam, build_losses_and_metrics, build_super_resolution_unet
from adunet_amd.parallel import DataParallel

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16"])
args = ap.parse_args()
dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
rng = np.random.default_rng(5)
steps = 4
per = 2
OVERFLOW_STEP, OVERFLOW_RANK = 1, 1
batches = []
for i in range(steps):
    hr = rng.random((per * world, 32, 32, 3), dtype=np.float32)
    lr = np.clip(hr + 0.05 * rng.standard_normal(hr.shape, dtype=np.float32), 0, 1).astype(np.float32)
    if dtype == torch.float16 and i == OVERFLOW_STEP:
        lr[OVERFLOW_RANK * per, 5, 7, 1] = 1e30           # lands in rank 1's half only
    batches.append((lr, hr))

def make():
    m, _ = build_super_resolution_unet(0.5, depth_override=2, input_size=32, dtype=dtype, device=dev)
    loss, metrics = build_losses_and_metrics("charbonnier")
    m.compile(optimizer=Adam(1e-3), loss=loss, metrics=metrics)
    m._require_device()
    m.set_weights(m.initial_weights(np.random.default_rng(1), head_uniform=0.05))
    return m

def scaler_state(m):
    sc = m._scaler()
    return sc.state.clone() if sc is not None else torch.zeros(8, device=dev)

results, g0 = {}, {}
for mode in ("eager", "graph"):
    m = make()
    DataParallel(m, bucket_bytes=1 << 18)
    sl = slice(rank * per, (rank + 1) * per)
    if mode == "graph":
        step = m.make_graphed_train_step(batches[0][0][sl], batches[0][1][sl], capture_only=True)
    else:
        step = m.train_on_batch
    for k, (lr, hr) in enumerate(batches):
        step(lr[sl], hr[sl])
        if k == 0:
            torch.cuda.synchronize()
            g0[mode] = m.G.clone()           # the all-reduced (summed) flat gradient of step 0, before any optimizer effect
    torch.cuda.synchronize()
    results[mode] = (m.P.clone(), scaler_state(m))
    index = m.index
    del m, step

# every rank must hold the same weights and the same scaler state (bitwise): compare through rank 0
ok = True
for mode, (p, st) in results.items():
    both_p = [torch.empty_like(p.cpu()) for _ in range(world)]
    both_s = [torch.empty_like(st.cpu()) for _ in range(world)]
    dist.all_gather(both_p, p.cpu())
    dist.all_gather(both_s, st.cpu())
    same = all(torch.equal(both_p[0], q) for q in both_p[1:]) and all(torch.equal(both_s[0], q) for q in both_s[1:])
    if rank == 0:
        print(f"{mode}: ranks hold identical weights and scaler state: {same}", flush=True)
    ok &= same
ref = None
if rank == 0:
    m = make()                               # one process, whole batch
    for k, (lr, hr) in enumerate(batches):
        m.train_on_batch(lr, hr)
        if k == 0:
            torch.cuda.synchronize()
            ref_g0 = m.G.clone()
    torch.cuda.synchronize()
    ref, ref_state = m.P.clone(), scaler_state(m)
    # The exchange itself, BEFORE Adam (which normalises magnitudes and would hide a dropped or mis-scaled bucket behind its
    # one-step bound -- ADVICE r04): the summed gradient of step 0 over the ranks / world against the single process's, per
    # parameter tensor, relative to that tensor's largest gradient.  fp32: summation order; 16-bit: stored roundings of
    # activations that differ between a 2-image and a 4-image launch (a dropped bucket reads 1.0, a missing 1 / world 1.0 too)
    gtol = 1e-4 if dtype == torch.float32 else 5e-2
    for mode, g in g0.items():
        worst, where = 0.0, None
        for name, (off, shape) in index.items():
            cnt = int(np.prod(shape))
            a, b = g[off:off + cnt] / world, ref_g0[off:off + cnt]
            e = float((a - b).abs().max() / (b.abs().max() + 1e-30))
            if e > worst:
                worst, where = e, name
        print(f"{mode}: step-0 gradient, worst tensor {where}: max |G_dp / world - G_single| / max|G_single| = {worst:.3e}", flush=True)
        ok &= worst < gtol
    # fp32: summation order of the two halves only.  16-bit: a rank's launches see 2 images, the single process 4 -- other tile
    # geometries / split factors, i.e. another fp32 accumulation order inside the convolutions, which flips stored 16-bit
    # roundings and cascades; where a gradient element is noise, Adam's g / sqrt(v) turns that into up to a whole
    # step (lr = 1e-3).  Bound: one step relative to the weights' scale (observed 3-4e-4 after four steps)
    tol = 1e-5 if dtype == torch.float32 else 1e-3
    for mode, (p, st) in results.items():
        err = float((p - ref).abs().max() / ref.abs().max())
        print(f"{mode}: max |P_dp - P_single| / max|P| = {err:.3e}", flush=True)
        ok &= err < tol and bool(torch.isfinite(p).all())
        if dtype == torch.float16:
            s = st.cpu().tolist()
            print(f"{mode}: loss scale {s[0]:.0f}, applied {int(s[4])}, skipped {int(s[5])}", flush=True)
            # one skipped step (the overflow on ONE rank), scale halved once, the other steps applied -- as the single process
            ok &= s[0] == 2.0 ** 14 and int(s[4]) == steps - 1 and int(s[5]) == 1 and torch.equal(st, ref_state)
    print("eager == graph bitwise:", bool(torch.equal(results["eager"][0], results["graph"][0])), flush=True)
    ok &= bool(torch.equal(results["eager"][0], results["graph"][0])) and bool(torch.equal(results["eager"][1], results["graph"][1]))
t = torch.tensor([1.0 if ok else 0.0])
dist.all_reduce(t, op=dist.ReduceOp.MIN)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if t.item() == 1.0 else 1)
