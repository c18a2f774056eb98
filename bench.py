#!/usr/bin/env python3
"""Headline benchmark: SR train images/sec @256x256, bf16, on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full Keras train step of the adaptive-depth SR U-Net (forward, Charbonnier loss, backward,
RCCL gradient all-reduce when N>1, Keras-form Adam, weight-operand repack) over one synthetic batch that is
already resident in HBM.  Workload K2' (SURVEY 8d): scale 0.25, depth 4, 256x256 patches, 64 images per
GPU (weak scaling), bf16 activations / fp32 accumulation / fp32 master weights.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

METRIC = "SR ×4 train images/sec @256×256 bf16, 1/2/4/8 MI355X; PSNR vs ref"
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
WORKLOADS = {
    # name: (scale, depth, patch, per-GPU batch)
    "K2p": (0.25, 4, 256, 64),
    "R3": (0.5, 3, 256, 64),
    "K2": (0.25, 4, 512, 16),
    "K1": (0.5, 2, 128, 4),
}


def conv_flops_per_image(model):
    """Algorithmic conv FLOPs (2*H*W*Cin*Cout*k^2, real Cin) per image: forward total and first conv."""
    fwd = first = 0.0
    for i, cs in enumerate(model.convs.values()):
        f = 2.0 * cs.hw * cs.hw * cs.cin * cs.cout * cs.k * cs.k
        fwd += f
        if i == 0:
            first = f
    return fwd, first


def synth_batch(rank, n, p, device):
    rng = np.random.default_rng(1234 + rank)
    hr = rng.random((n, p, p, 3), dtype=np.float32)
    lr = np.clip(hr + 0.05 * rng.standard_normal(hr.shape, dtype=np.float32), 0, 1).astype(np.float32)
    return torch.from_numpy(lr).to(device), torch.from_numpy(hr).to(device)


def pmc_traffic(workload, dtype, batch):
    """HBM bytes per launch of the conv3x3_fwd family from the committed rocprofv3 PMC passes (profiles/README.md);
    None when the run is not the configuration those passes were taken on."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if (workload, dtype, batch) != ("K2p", "bf16", 64) or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)["families"]["conv3x3_fwd"]["hbm_bytes_per_launch"]


def cpu_baseline(scale, depth, patch, budget_seconds=12.0):
    """Oracle ("port") timed on the host cores: whole fp32 train steps of batch 1 until `budget_seconds` have passed."""
    from oracle.sr_unet import SRUNetOracle
    try:
        from threadpoolctl import threadpool_info
        cores = max([d.get("num_threads", 1) for d in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        cores = os.cpu_count() or 1
    rng = np.random.default_rng(1234)
    m = SRUNetOracle(scale, depth, patch)
    params = m.init_params(rng, dtype=np.float32, head_uniform=0.05)
    state = {}
    hr = rng.random((1, patch, patch, 3), dtype=np.float32)
    lr = np.clip(hr + 0.05 * rng.standard_normal(hr.shape, dtype=np.float32), 0, 1).astype(np.float32)
    t0 = time.time()
    budget_images = 0
    while budget_images == 0 or time.time() - t0 < budget_seconds:
        m.train_step(params, state, lr, hr, lr=1e-4)
        budget_images += 1
    dt = time.time() - t0
    return {"value": budget_images / dt, "unit": "images/s", "cores": int(cores), "kind": "port",
            "sample": f"{budget_images} train step(s) of batch 1 on the same model (NumPy float32 oracle, {dt:.1f} s)"}


def launch_ranks(n: int) -> int:
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same flags>` as a child process."""
    import socket
    import subprocess
    with socket.socket() as sock:          # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="K2p", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print the per-op-family time table to stderr")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves.  Nothing in this process has touched the GPU
        # yet (no HIP call, no torch.cuda query), the ranks are fresh child processes (not an exec of this one), rank 0
        # prints the JSON line on the inherited stdout and the children's return code becomes ours.
        raise SystemExit(launch_ranks(args.gpus))
    if world != args.gpus:
        args.gpus = world
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import torch.distributed as dist
    use_dist = "RANK" in os.environ          # launched by torch.distributed.run (also exercised at world size 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from adunet_amd import ops
    from adunet_amd.model import Adam, build_losses_and_metrics, build_super_resolution_unet
    from adunet_amd.parallel import DataParallel

    scale, depth, patch, batch = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model, info = build_super_resolution_unet(scale, depth_override=depth, input_size=patch, dtype=dtype, device=device)
    loss, metrics = build_losses_and_metrics("charbonnier")
    model.compile(optimizer=Adam(learning_rate=1e-4), loss=loss, metrics=metrics, jit_compile=False)
    model._require_device()
    # non-degenerate head so that every gradient is exercised (SURVEY 8d workload recipe)
    model.set_weights(model.initial_weights(np.random.default_rng(1234), head_uniform=0.05))
    if use_dist:
        DataParallel(model)
    lr_img, hr_img = synth_batch(rank, batch, patch, device)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The ~190 launches of a step are captured once in hipGraphs and replayed (no tracing compiler: the same
    # hand-written kernels, minus the Python launch loop).  Data parallel: the capture is cut where a gradient bucket
    # becomes final and the RCCL all-reduces are launched eagerly between the segments on their own stream.
    graphed = not args.eager
    step_fn = model.make_graphed_train_step(lr_img, hr_img) if graphed else model.train_on_batch
    for _ in range(args.warmup):
        step_fn(lr_img, hr_img)
    timer = ops.KernelTimer()
    sync()
    if not graphed:
        ops.set_timer(timer)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last_loss, last_psnr = step_fn(lr_img, hr_img)
    sync()
    elapsed = time.perf_counter() - t0
    ops.set_timer(None)
    if graphed:
        # per-kernel HIP-event timing needs eager launches: the same step, instrumented, right after the timed region
        ops.set_timer(timer)
        for _ in range(min(args.steps, 3)):
            last_loss, last_psnr = model.train_on_batch(lr_img, hr_img)
        torch.cuda.synchronize()
        ops.set_timer(None)
    timed_steps = min(args.steps, 3) if graphed else args.steps
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    if rank == 0:
        fwd, first = conv_flops_per_image(model)
        f_step = 3.0 * fwd - first
        summ = timer.summary()
        # dominant kernel family: the forward convs and the dgrads (same kernels on the rotated weight pack) that run
        # WITHOUT the fused LayerNorm epilogue; the five fused launches are their own timer family.  FLOPs are the
        # algorithmic ones: what the wrappers launched minus the zero-padded channels of the first conv (3 -> 32).
        n_launch, ms = summ["conv3x3_fwd"]
        padded_first = first * (model._cin_pad(next(iter(model.convs.values()))) / 3.0 - 1.0) * batch * timed_steps
        if "conv3x3_c3_ln_relu_fwd" in summ:      # bf16: the first conv has its own 3-channel kernel, nothing was padded
            padded_first = 0.0
        flops_kernel = timer.work("conv3x3_fwd") - padded_first
        achieved = flops_kernel / (ms * 1e-3) / 1e12
        fused = summ.get("conv3x3_ln_relu_fwd", (0, 0.0))
        total_ms = sum(v[1] for v in summ.values())
        if args.breakdown:
            print(f"{'op family':<22}{'launches/step':>14}{'ms/step':>10}{'share':>8}", file=sys.stderr)
            for k, (cnt, t_ms) in sorted(summ.items(), key=lambda kv: -kv[1][1]):
                print(f"{k:<22}{cnt / timed_steps:>14.1f}{t_ms / timed_steps:>10.3f}{t_ms / total_ms:>8.1%}", file=sys.stderr)
            print(f"{'(sum of op events)':<22}{'':>14}{total_ms / timed_steps:>10.3f}", file=sys.stderr)
        img_s = batch * world * args.steps / elapsed
        line = {
            "metric": METRIC, "value": img_s, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1000.0 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: SR U-Net scale {scale} depth {depth} patch {patch} train step",
                       "global_batch": batch * world, "per_gpu_batch": batch, "params": model.count_params(),
                       "parallelism": f"dp{world}", "launch": "hipGraph replay" if graphed else "eager", "conv_gflop_per_image_step": f_step / 1e9,
                       "model_tflops": img_s * f_step / 1e12, "final_loss": float(last_loss),
                       "final_psnr": float(last_psnr)},
            "roofline": {"bound": "mfma",
                         "kernel": "forward-conv / dgrad family: conv3x3_fwd_wres_kernel, conv3x3_fwd_ws_kernel, "
                                   "conv3x3_fwd_kernel (+ splitk_finalize_kernel), launches without the fused "
                                   "LayerNorm epilogue",
                         "fused_ln_launches_per_step": fused[0] / timed_steps, "fused_ln_ms_per_step": fused[1] / timed_steps,
                         "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS,
                         "traffic": pmc_traffic(args.workload, args.dtype, batch),
                         "algorithmic_bytes_per_launch": timer.nbytes("conv3x3_fwd") / n_launch,
                         "launches_per_step": n_launch / timed_steps, "avg_launch_ms": ms / n_launch,
                         "timing": ("HIP events around every launch of %d eager steps run right after the graph-replayed "
                                    "timed region" % timed_steps) if graphed else "HIP events inside the timed region",
                         "gflop_per_launch": flops_kernel / n_launch / 1e9},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scale, depth, patch)
        print(json.dumps(line), flush=True)
    if use_dist:
        del step_fn, model            # graphs and side streams go before the communicator they reference
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
