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
    # the reference's own Experiment-2 shapes (Super_resolution/sbatch_scripts/run_experiment_adaptive_depth.sh:36-66)
    "E2s06": (0.6, 4, 256, 32),     # 256/154/93/56/34
    "E2s07": (0.7, 5, 256, 8),      # 256/180/126/89/63/45, 2048-channel bottleneck
}
# BASELINE config 5 ("adaptive per-sample depth 2-6, mixed SR + seg multitask, fp16"): BUILD-DEFINED semantics (the reference has
# no such path: adunet_amd/multitask.py).  One "step" = one pass over this fixed mixed stream, one captured graph per item:
# the Experiment-2 rows 0.3 -> 2, 0.5 -> 3, 0.6 -> 4, 0.7 -> 5 at their bench batches and the segmentation task.
K5_STREAM = [("sr", 0.3, 64), ("sr", 0.5, 64), ("sr", 0.6, 32), ("sr", 0.7, 8), ("seg", None, 16)]
# BASELINE config 3 as the reference's source defines it (ISIC binary masks at 256 x 256, not Cityscapes):
# Segmenation/code/train_adaptive_unet.py:335-362 build_adaptive_depth_unet(256, 64, depth) -- Conv3x3 + BatchNorm + ReLU x 2,
# MaxPool2, bilinear x2, concat, sigmoid head, protocol A/B loss -- and unet_vinillia.py:72-91 build_unet (LayerNorm, Conv2DTranspose)
SEG_WORKLOADS = {
    # name: (builder, depth, base channels, per-GPU batch, protocol)
    "K3": ("bn", 5, 64, 8, "A"),            # build_adaptive_depth_unet(256, 64, 5), protocol A batch 8 (:382-392)
    "K3d4": ("bn", 4, 64, 16, "B"),         # the reference's default depth 4 (:55-56), protocol B batch 16 (:393-403)
    "K3ln": ("ln", 4, 64, 16, "B"),         # vanilla baseline build_unet at 64 base channels
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


def kernel_source_stamp():
    """sha256 of the kernel sources a PMC measurement belongs to: a committed traffic figure is only quoted while the
    kernels it was measured on are the ones in the tree."""
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "adaptive-depth-u-net-for-image-super-resolution-segmentation_amd")
    # every kernel source of the step, plus the build recipe (compiler flags change the traffic of the same source)
    for name in ("csrc/common.h", "csrc/conv.hip", "csrc/upconv.hip", "csrc/norm.hip", "csrc/resize.hip", "csrc/head.hip",
                 "csrc/optim.hip", "build.py"):
        with open(os.path.join(pkg, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_record(workload, dtype, batch):
    """The committed rocprofv3 PMC passes (profiles/README.md) of the newest round, or None when they were measured on
    another configuration or on other kernel sources (stale).  PMC counters need their own rocprofv3 runs, so the figures
    cannot be taken by the run that prints them; the sha256 stamp ties them to the sources in the tree."""
    import glob
    import re
    cands = glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))
    cands.sort(key=lambda p: (int(re.search(r"r(\d+)\w*_pmc_traffic", os.path.basename(p)).group(1)), p))   # r10 after r02
    if (workload, dtype, batch) != ("K2p", "bf16", 64) or not cands:
        return None
    with open(cands[-1]) as f:
        rec = json.load(f)
    return rec if rec.get("kernel_source_stamp") == kernel_source_stamp() else None


def cpu_baseline(scale, depth, patch, workload):
    """The CPU path timed beside the kernels on this box's host cores (rank 0, N = 1 only), ~35 s in total.

    TensorFlow/Keras cannot run here, so the baseline is a port: `value` = whole train steps of the SAME
    workload at batch 4 on the PyTorch-CPU stand-in (oracle/torch_standin.py: F.conv2d/oneDNN + layer_norm +
    interpolate(antialias) + autograd, float32, all cores); next to it the reference's own CPU-runnable shapes K1 and R3
    at the reference CLI's batch 4 and the NumPy oracle the parity tests use (batch 1)."""
    from oracle.sr_unet import SRUNetOracle
    from oracle.torch_standin import time_train_steps
    ips, steps, dt, threads = time_train_steps(scale, depth, patch, 4, 10.0)
    out = {"value": ips, "unit": "images/s", "cores": int(threads), "kind": "port",
           "cores_note": "threads = this process's CPU share (cgroup quota), not the logical CPUs the box shows",
           "implementation": "PyTorch-CPU float32 stand-in for the TF/Keras CPU path (TF is not installable here)",
           "sample": f"{steps} train step(s) of batch 4 of {workload} after one warm-up step ({dt:.1f} s)"}
    for name, (sc, dp, pp) in (("K1", (0.5, 2, 128)), ("R3", (0.5, 3, 256))):
        v, st, d, _ = time_train_steps(sc, dp, pp, 4, 6.0)
        out[f"{name}_batch4_torch_cpu"] = {"value": v, "unit": "images/s", "sample": f"{st} train step(s), {d:.1f} s"}
    from oracle.ops import cpu_share
    try:                                   # OpenBLAS sized to the CPU share as well (it would start one thread per visible CPU)
        from threadpoolctl import threadpool_limits
        blas_limit = threadpool_limits(limits=cpu_share(cap=1 << 10))
    except Exception:
        blas_limit = None
    rng = np.random.default_rng(1234)
    m = SRUNetOracle(scale, depth, patch)
    params = m.init_params(rng, dtype=np.float32, head_uniform=0.05)
    state = {}
    hr = rng.random((1, patch, patch, 3), dtype=np.float32)
    lr = np.clip(hr + 0.05 * rng.standard_normal(hr.shape, dtype=np.float32), 0, 1).astype(np.float32)
    t0 = time.time()
    n = 0
    while n == 0 or time.time() - t0 < 6.0:
        m.train_step(params, state, lr, hr, lr=1e-4)
        n += 1
    d = time.time() - t0
    out["numpy_oracle"] = {"value": n / d, "unit": "images/s", "sample": f"{n} train step(s) of batch 1 of {workload}, float32, {d:.1f} s"}
    del blas_limit
    return out


def micro_kernel(device, iters=50):
    """north_star's kernel target, measured in this process right after the timed region: ONE 3x3 convolution 64 -> 64
    on 32 x 256 x 256 bf16 (SURVEY 8d row mu), forward / dgrad / wgrad each timed with HIP events on the launch stream
    over `iters` back-to-back launches (after 150 warm-up launches), and summed: 3 x 154.6 GFLOP over the three times."""
    from adunet_amd import ops
    n, hw, c = 32, 256, 64
    g = torch.Generator(device="cpu").manual_seed(7)
    x = (torch.rand((n, hw, hw, c), generator=g) * 2 - 1).to(device=device, dtype=torch.bfloat16)
    dz = (torch.rand((n, hw, hw, c), generator=g) * 2 - 1).to(device=device, dtype=torch.bfloat16)
    w = ((torch.rand((3, 3, c, c), generator=g) * 2 - 1) * 0.05).to(device)
    bias = torch.zeros(c, device=device)
    wf, wd = ops.conv3x3_pack(w, c, torch.bfloat16)
    dw = torch.empty_like(w)
    ws = ops.Workspace(device)
    flops = 2.0 * n * hw * hw * 9 * c * c
    jobs = {"fwd": lambda: ops.conv3x3_fwd(x, None, wf, bias, c),
            "dgrad": lambda: ops.conv3x3_fwd(dz, None, wd, None, c),
            "wgrad": lambda: ops.conv3x3_wgrad(x, None, dz, dw, c, ws)}
    out, total_ms = {"shape": f"N={n}, {hw}x{hw}, {c}->{c}, bf16", "gflop_per_pass": flops / 1e9, "iters": iters}, 0.0
    for name, fn in jobs.items():
        # the chip lowers its clock under sustained MFMA load (MI355X_MICROARCH.md, DVFS): a pass timed right after an
        # idle gap reads ~15 % high, so every pass is timed after ~25 ms of its own back-to-back launches
        for _ in range(150):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        total_ms += ms
        out[name] = {"ms": ms, "tflops": flops / ms / 1e9, "frac": flops / ms / 1e9 / PEAK_BF16_TFLOPS}
    out["fwd_dgrad_wgrad"] = {"ms": total_ms, "tflops": 3 * flops / total_ms / 1e9,
                              "frac": 3 * flops / total_ms / 1e9 / PEAK_BF16_TFLOPS}
    return out


def inkernel_clocks():
    """Secondary figure only: the shader clock the dense conv launches really run at, measured INSIDE the
    kernels of a diagnostic build (tools/inkernel_clock.py: s_memtime over s_memrealtime around MFMA wave 0 of every workgroup
    after >= 2 s of back-to-back launches on random data) and committed under profiles/.  The chip lowers its clock under MFMA
    load, so `frac` (against 2.5 PFLOP/s = 2.4 GHz, what the roofline is graded on) understates how close the matrix pipes are
    to what the clock allows; `frac_at_clock` = TFLOP/s / (2.5 PF x clock / 2.4 GHz).  Replaces r03's rocm-smi power model."""
    import glob
    import re
    cands = glob.glob(os.path.join(ROOT, "profiles", "r*_inkernel_clock.json"))
    cands.sort(key=lambda p: (int(re.search(r"r(\d+)", os.path.basename(p)).group(1)), p))
    if not cands:
        return None, None
    with open(cands[-1]) as f:
        rec = json.load(f)
    fam_kernel = {"fwd_dgrad": "conv3x3_fwd_wres_kernel<.,0>", "fused_ln_fwd": "conv3x3_fwd_wres_kernel<.,2>",
                  "dgrad_ln_bwd_fused": "conv3x3_fwd_wres_kernel<.,4>", "wgrad": "conv3x3_wgrad_ws_kernel"}
    return ({f: rec["kernels"][k]["clock_ghz_median"] for f, k in fam_kernel.items() if k in rec.get("kernels", {})},
            "profiles/" + os.path.basename(cands[-1]))


def run_k5(args, rank, world, device, use_dist):
    """BASELINE config 5 as adunet_amd.multitask defines it (build-defined: the reference trains one independent model per
    (scale, depth), run_experiment_adaptive_depth.sh:47-55, and segmentation in a separate script).  A step is one pass over
    K5_STREAM: every item is one graph-replayed train step of its own model (own weights, own Adam, under f16 its own dynamic
    loss scaler); under torch.distributed every model has its own bucketed gradient exchange.  Reports images/s in total and
    per item (HIP events around each item's replay inside the timed region)."""
    import torch.distributed as dist
    from adunet_amd import multitask as M
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    patch = 256
    bank = M.AdaptiveDepthBank(input_size=patch, dtype=dtype, device=device, learning_rate=1e-4, seg_depth=4)
    if use_dist:
        bank.data_parallel()
    rng = np.random.default_rng(1234 + rank)
    items = []
    for task, scale, nb in K5_STREAM:
        if args.batch:
            nb = args.batch
        if task == "sr":
            hr = rng.random((nb, patch, patch, 3), dtype=np.float32)
            lr = np.clip(hr + 0.05 * rng.standard_normal(hr.shape, dtype=np.float32), 0, 1).astype(np.float32)
            items.append(("sr", scale, torch.from_numpy(lr).to(device), torch.from_numpy(hr).to(device)))
        else:
            img = rng.random((nb, patch, patch, 3), dtype=np.float32)
            mask = (rng.random((nb, patch, patch, 1)) < 0.3).astype(np.float32)
            items.append(("seg", torch.from_numpy(img).to(device), torch.from_numpy(mask).to(device)))

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def one_pass(events=None):
        last = []
        for i, it in enumerate(items):
            if events is not None:
                events[i][0].record()
            last.append(bank.train_on_batch(it[0], *it[1:], graphed=True))
            if events is not None:
                events[i][1].record()
        return last

    for _ in range(max(args.warmup, 1)):       # the first pass builds the models and captures the graphs
        one_pass()
    sync()
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in items] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        last = one_pass(ev[k])
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    per_pass = sum(it[-1].shape[0] for it in items)
    img_s = per_pass * world * args.steps / elapsed
    per_item, flop_pass = [], 0.0
    for i, ((task, scale, _), it) in enumerate(zip(K5_STREAM, items)):
        nb = it[-1].shape[0]
        ms = sum(e[i][0].elapsed_time(e[i][1]) for e in ev) / args.steps
        model = bank.sr[(scale, M.route_depth(scale, input_size=patch))] if task == "sr" else bank.seg
        fwd, first = conv_flops_per_image(model)
        f_step = 3.0 * fwd - first
        flop_pass += f_step * nb
        per_item.append({"task": task, "scale": scale, "depth": (model.depth if task == "sr" else bank.seg_depth), "batch": nb,
                         "params": model.count_params(), "ms_per_step": ms, "images_per_s": nb / ms * 1e3,
                         "conv_gflop_per_image_step": f_step / 1e9,
                         "frac_step": nb / ms * 1e3 * f_step / 1e12 / PEAK_BF16_TFLOPS,
                         "final_loss": float(last[i][1])})
    ms_step = elapsed * 1e3 / args.steps
    tfl = flop_pass / (ms_step * 1e-3) / 1e12
    line = {"metric": METRIC, "value": img_s, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "K5: BUILD-DEFINED mixed stream of BASELINE config 5 (no reference semantics: "
                                   "adunet_amd/multitask.py) -- SR scale/depth 0.3/2, 0.5/3, 0.6/4, 0.7/5 + segmentation depth 4, "
                                   "patch 256, one model and one captured graph per item; a step = one pass over the stream",
                       "global_batch": per_pass * world, "per_gpu_batch": per_pass, "parallelism": f"dp{world}",
                       "launch": "hipGraph replay per item", "items": per_item},
            "roofline": {"bound": "mfma", "achieved": tfl, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": tfl / PEAK_BF16_TFLOPS, "traffic": None,
                         "what": "reference-graph conv FLOPs of the whole pass (3 F_fwd - F_first per image and model) over the "
                                 "pass time: the step-level figure (`frac_step` of the single-model workloads); per item in config.items"}}
    if use_dist:
        line["ranks"] = dist.get_world_size()
        line["dist_backend"] = args.backend
        if args.backend == "nccl":
            line["rccl_ranks"] = dist.get_world_size()
        torch.cuda.synchronize()
        bank.close()
    return line


def run_seg(args, rank, world, device, use_dist):
    """BASELINE config 3 (restated by SURVEY 8d as K3): one train step of the segmentation U-Net -- forward, BCE + Dice loss,
    backward, Keras Adam, operand repack -- graph-replayed on a resident synthetic batch (image U[0,1), mask Bernoulli(0.3)).
    Same line as the SR workloads: whole-step `frac_step` on the reference graph's conv FLOPs, the MFMA families, every
    HBM-bound op with its algorithmic bytes and rate.  BatchNorm statistics are per replica."""
    import torch.distributed as dist
    from adunet_amd import ops
    from adunet_amd import seg_model as S
    from adunet_amd.parallel import DataParallel
    kind, depth, base, batch, proto_name = SEG_WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    patch = 256
    model = (S.build_adaptive_depth_unet(patch, base, depth, dtype=dtype, device=device) if kind == "bn"
             else S.build_unet(patch, 1, base, depth, dtype=dtype, device=device))
    proto = S.PROTOCOLS[proto_name]
    model.compile(optimizer=S.build_optimizer(proto, steps_per_epoch=1000, epochs=100), loss=proto.loss_builder())
    model._require_device()
    if use_dist:
        DataParallel(model)
    rng = np.random.default_rng(1234 + rank)
    img = torch.from_numpy(rng.random((batch, patch, patch, 3), dtype=np.float32)).to(device)
    mask = torch.from_numpy((rng.random((batch, patch, patch, 1)) < 0.3).astype(np.float32)).to(device)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    graphed = not args.eager
    step_fn = model.make_graphed_train_step(img, mask) if graphed else model.train_on_batch
    for _ in range(args.warmup):
        step_fn(img, mask)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step_fn(img, mask)
    sync()
    elapsed = time.perf_counter() - t0
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    timed_steps = min(args.steps, 3)
    for _ in range(timed_steps):
        last = model.train_on_batch(img, mask)
    torch.cuda.synchronize()
    ops.set_timer(None)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    if rank != 0:
        return None
    fwd, first = conv_flops_per_image(model)
    f_step = 3.0 * fwd - first
    summ = timer.summary()
    per_step = 1.0 / timed_steps
    total_ms = sum(v[1] for v in summ.values()) * per_step
    pad_first = first * (model._cin_pad(next(iter(model.convs.values()))) / 3.0 - 1.0) * batch     # the 3-channel input is zero-padded

    def family(names, minus=0.0):
        cnt = sum(summ.get(k, (0, 0.0))[0] for k in names)
        ms = sum(summ.get(k, (0, 0.0))[1] for k in names) * per_step
        fl = sum(timer.work(k) for k in names) * per_step - minus
        nb = sum(timer.nbytes(k) for k in names) * per_step
        return {"launches_per_step": cnt * per_step, "gflop_per_step": fl / 1e9, "ms_per_step": ms,
                "tflops": fl / ms / 1e9 if ms > 0 else None, "frac": fl / ms / 1e9 / PEAK_BF16_TFLOPS if ms > 0 else None,
                "algorithmic_gb_per_step": nb / 1e9, "hbm_tb_per_s": nb / ms / 1e9 if ms > 0 else None,
                "flop_per_byte": fl / nb if nb > 0 else None, "share_of_step": ms / total_ms if total_ms > 0 else None}

    fam = {"fwd_dgrad": family(["conv3x3_fwd"], pad_first),
           "fused_ln_fwd": family(["conv3x3_ln_relu_fwd"]),
           "fused_bn_fwd": family(["conv3x3_bn_stats_fwd"], pad_first),
           "dgrad_bn_bwd_fused": family(["conv3x3_dgrad_bn_bwd"]),
           "wgrad": family(["conv3x3_wgrad"], pad_first)}
    fam = {k: v for k, v in fam.items() if v["launches_per_step"] > 0}
    conv_ms = sum(f["ms_per_step"] for f in fam.values())
    conv_gf = sum(f["gflop_per_step"] for f in fam.values())
    hbm_ops = {}
    for k, (cnt, t_ms) in sorted(summ.items(), key=lambda kv: -kv[1][1]):
        if k.startswith("conv3x3_") and k != "conv3x3_pack":
            continue
        nb = timer.nbytes(k)
        hbm_ops[k] = {"launches_per_step": cnt * per_step, "ms_per_step": t_ms * per_step,
                      "algorithmic_gb_per_step": nb * per_step / 1e9 if nb else None,
                      "tb_per_s": nb / t_ms / 1e9 if nb else None}
    if args.breakdown:
        print(f"{'op family':<26}{'launches/step':>14}{'ms/step':>10}{'share':>8}{'TB/s':>8}", file=sys.stderr)
        for k, (cnt, t_ms) in sorted(summ.items(), key=lambda kv: -kv[1][1]):
            nb = timer.nbytes(k)
            print(f"{k:<26}{cnt * per_step:>14.1f}{t_ms * per_step:>10.3f}{t_ms * per_step / total_ms:>8.1%}"
                  f"{(nb / t_ms / 1e9 if nb else float('nan')):>8.2f}", file=sys.stderr)
        print(f"{'(sum of op events)':<26}{'':>14}{total_ms:>10.3f}", file=sys.stderr)
    dom_name = max(fam, key=lambda k: fam[k]["ms_per_step"])
    dom = fam[dom_name]
    img_s = batch * world * args.steps / elapsed
    ms_step = elapsed * 1e3 / args.steps
    loss_v, dice_v, iou_v = (float(v) for v in last)
    line = {"metric": METRIC, "value": img_s, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: segmentation U-Net {model.name} (BASELINE config 3 as the reference's source "
                                   f"defines it: 256 x 256 binary masks), protocol {proto_name} train step",
                       "global_batch": batch * world, "per_gpu_batch": batch, "params": model.count_params(),
                       "parallelism": f"dp{world}", "launch": "hipGraph replay" if graphed else "eager",
                       "conv_gflop_per_image_step": f_step / 1e9, "final_loss": loss_v, "final_dice": dice_v, "final_iou": iou_v},
            "roofline": {"bound": "mfma", "family": dom_name, "achieved": dom["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": dom["frac"], "traffic": None,
                         "frac_step": img_s / world * f_step / 1e12 / PEAK_BF16_TFLOPS,
                         "algorithmic_gflop_per_step": f_step * batch / 1e9, "executed_gflop_per_step": conv_gf,
                         "frac_all_conv_kernels": conv_gf / conv_ms / PEAK_BF16_TFLOPS if conv_ms > 0 else None,
                         "families": fam, "hbm_ops": hbm_ops, "non_conv_ms_per_step": total_ms - conv_ms,
                         "timing": "HIP events around every launch of %d eager steps run right after the graph-replayed timed "
                                   "region" % timed_steps}}
    if use_dist:
        line["ranks"] = dist.get_world_size()
        line["dist_backend"] = args.backend
    return line


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
    ap.add_argument("--workload", default="K2p", choices=sorted(WORKLOADS) + sorted(SEG_WORKLOADS) + ["K5"],
                    help="K5 = BASELINE config 5, build-defined: a fixed mixed SR (depths 2-5) + segmentation stream, default --dtype f16")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override")
    ap.add_argument("--dtype", default=None, choices=["bf16", "f16", "f32"],
                    help="f16 = the reference's mixed_float16 policy (fp16 kernels + dynamic loss scaling)")
    ap.add_argument("--feed", default="resident", choices=["resident", "loader"],
                    help="loader: after the resident-batch timed region (which stays `value`), time the same graph-replayed step fed "
                         "by the host data path -- PrefetchPatchLoader crop workers on a synthetic in-memory uint8 image set, uint8 "
                         "crops over PCIe, DeviceDegrader LR synthesis in HBM (shared/pipeline.py:214-246's contract) -- and report it "
                         "as `feed`")
    ap.add_argument("--feed-workers", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-micro", action="store_true", help="skip the 64->64 @256x256 N=32 micro-kernel measurement")
    ap.add_argument("--breakdown", action="store_true", help="print the per-op-family time table to stderr")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N > 1 run: nccl = RCCL over xGMI, one rank per GPU (the driver's "
                         "scaling run); gloo = a rehearsal of the same rank logic on a box with fewer GPUs than ranks (the "
                         "ranks share the visible GPUs, gradients travel through host memory: NOT a scaling measurement)")
    args = ap.parse_args()
    if args.dtype is None:
        args.dtype = "f16" if args.workload == "K5" else "bf16"

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
    feed_ds = None
    if args.feed == "loader" and args.workload != "K5":
        # the crop workers are FORKED: start them before this process touches the GPU (they only ever run NumPy)
        from adunet_amd.pipeline import FastFeedDataset
        f_scale, _, f_patch, f_batch = WORKLOADS[args.workload]
        rng_img = np.random.default_rng(99 + rank)
        images = [rng_img.integers(0, 256, (2 * f_patch, 2 * f_patch, 3), dtype=np.uint8) for _ in range(48)]
        feed_ds = FastFeedDataset(images, f_patch, args.batch or f_batch, f_scale, patches_per_image=4, seed=1234,
                                  workers=args.feed_workers, shard=(rank, world))
    # stdout carries exactly ONE line, the JSON record: libraries that write to file descriptor 1 (RCCL prints a version
    # banner there when its first communicator is created) are pointed at stderr for the rest of the run
    sys.stdout.flush()
    record_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    # one rank per GPU; under --backend gloo the ranks may outnumber the GPUs and share them round-robin
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    import torch.distributed as dist
    use_dist = "RANK" in os.environ          # launched by torch.distributed.run (also exercised at world size 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from adunet_amd import ops
    from adunet_amd.model import Adam, build_losses_and_metrics, build_super_resolution_unet
    from adunet_amd.parallel import DataParallel

    if args.workload == "K5" or args.workload in SEG_WORKLOADS:
        line = run_k5(args, rank, world, device, use_dist) if args.workload == "K5" else run_seg(args, rank, world, device, use_dist)
        if rank == 0:
            print(json.dumps(line), file=record_out, flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

    scale, depth, patch, batch = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    model, info = build_super_resolution_unet(scale, depth_override=depth, input_size=patch, dtype=dtype, device=device)
    loss, metrics = build_losses_and_metrics("charbonnier")
    model.compile(optimizer=Adam(learning_rate=1e-4), loss=loss, metrics=metrics, jit_compile=False)
    model._require_device()
    # non-degenerate head so that every gradient is exercised (SURVEY 8d workload recipe)
    model.set_weights(model.initial_weights(np.random.default_rng(1234), head_uniform=0.05))
    if use_dist:
        DataParallel(model)            # ADUNET_NATIVE_RCCL=1: bucket all-reduces through the library's ad_allreduce_bucket
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

    feed = None
    if feed_ds is not None:
        # the same step, fed by the host data path instead of one resident batch: crop workers -> shared-memory ring -> uint8
        # over PCIe on a copy stream (one batch ahead) -> LR synthesis in HBM -> copy into the graph's static inputs -> replay
        feed_ds.device = device
        it = iter(feed_ds)
        for _ in range(max(args.warmup, 2)):
            step_fn(*next(it))
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn(*next(it))
        sync()
        feed_elapsed = time.perf_counter() - t0
        # the loader alone (no GPU work): what the crop workers and the ring sustain
        t0 = time.perf_counter()
        for _ in range(args.steps):
            next(feed_ds.loader)
        loader_elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([feed_elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            feed_elapsed = float(t[0])
        feed = {"images_per_s": batch * world * args.steps / feed_elapsed, "ms_per_step": feed_elapsed * 1e3 / args.steps,
                "fraction_of_resident_rate": (elapsed / feed_elapsed),
                "h2d_bytes_per_step": feed_ds.h2d_bytes_per_batch, "host_ring_page_locked": bool(feed_ds.pinned),
                "crop_workers": args.feed_workers, "loader_alone_images_per_s": batch * args.steps / loader_elapsed,
                "path": "PrefetchPatchLoader (48 synthetic 512x512 uint8 images, forked crop workers, shared-memory ring) -> uint8 "
                        "crops over PCIe on a copy stream, one batch ahead -> DeviceDegrader (INTER_AREA shrink + INTER_CUBIC "
                        "enlarge as two ad_resample launches) -> graph replay"}
        del it
        feed_ds.close()

    exposed_ms = None
    if use_dist:
        # exposed gradient exchange: time the compute stream sits waiting for RCCL, measured over a few extra steps
        model._dp.measure = True
        probe = min(args.steps, 5)
        for _ in range(probe):
            step_fn(lr_img, hr_img)
        torch.cuda.synchronize()
        exposed_ms = model._dp.exposed_ms() / probe
        model._dp.measure = False

    if rank == 0:
        fwd, first = conv_flops_per_image(model)
        f_step = 3.0 * fwd - first
        summ = timer.summary()
        per_step = 1.0 / timed_steps
        c3 = "conv3x3_c3_ln_relu_fwd" in summ       # bf16: the first conv has its own 3-channel kernels, nothing is padded
        # the wrappers book what they launched; the first conv of the fp32 path is zero-padded 3 -> 16 channels and
        # those padded FLOPs are not algorithmic work
        pad_ratio = 0.0 if c3 else model._cin_pad(next(iter(model.convs.values()))) / 3.0 - 1.0
        padded_first = first * pad_ratio * batch

        def family(names, minus=0.0):
            cnt = sum(summ.get(k, (0, 0.0))[0] for k in names)
            ms = sum(summ.get(k, (0, 0.0))[1] for k in names) * per_step
            fl = sum(timer.work(k) for k in names) * per_step - minus
            nb = sum(timer.nbytes(k) for k in names) * per_step
            # both roofs for every family: MFMA (TFLOP/s, frac of the dense bf16 peak) and HBM (algorithmic bytes -- each
            # operand tensor once -- over the same time; frac of 8 TB/s).  A family whose `hbm_frac` sits near the ~0.55 a
            # mixed read / write stream reaches on this part (6.3 TB/s float4 copy = 0.79) is on the HBM roof, whatever its
            # MFMA fraction says
            return {"launches_per_step": cnt * per_step, "gflop_per_step": fl / 1e9, "ms_per_step": ms,
                    "tflops": fl / ms / 1e9 if ms > 0 else None,
                    "frac": fl / ms / 1e9 / PEAK_BF16_TFLOPS if ms > 0 else None,
                    "algorithmic_gb_per_step": nb / 1e9, "hbm_tb_per_s": nb / ms / 1e9 if ms > 0 else None,
                    "hbm_frac": nb / ms / 1e9 / 8.0 if ms > 0 else None,
                    "flop_per_byte": fl / nb if nb > 0 else None}

        fam = {
            # forward convs and dgrads (the same kernels on the rotated pack) launched WITHOUT the fused LayerNorm epilogue
            # (fp32 runs every Conv2D -> LayerNorm link as two launches, so the padded first conv is booked here)
            "fwd_dgrad": family(["conv3x3_fwd"], padded_first if args.dtype == "f32" else 0.0),
            # Conv2D -> LayerNorm -> ReLU in one launch (incl. the dedicated 3-channel first layer)
            "fused_ln_fwd": family(["conv3x3_ln_relu_fwd", "conv3x3_c3_ln_relu_fwd"],
                                   padded_first if args.dtype != "f32" else 0.0),
            "wgrad": family(["conv3x3_wgrad", "conv3x3_c3_wgrad"], padded_first),
            # dgrad launches that also apply the up-conv's ReLU gradient and sum its bias gradient in their epilogue (an extra
            # 128 B/pixel stream through the MFMA waves; replaces a separate relu_bwd pass)
            "dgrad_relu_fused": family(["conv3x3_dgrad_relu"]),
            # dgrad launches whose epilogue is the LayerNorm / ReLU backward of the layer below (z, mean, rstd streamed in,
            # ~900 VALU instructions per tile and wave; replaces a separate 1.6 GB layernorm_relu_bwd pass)
            "dgrad_ln_bwd_fused": family(["conv3x3_dgrad_ln_bwd"]),
            # the decoder's up-resize -> Conv3x3 pairs in their factored form (csrc/upconv.hip): the 1x1 bank GEMMs at the LOW
            # resolution (Y = x B, dx = dY B^T, dB = x^T dY).  GFLOP = what these launches EXECUTE (1 / ratio^2 of the 3x3
            # convolution they stand for); the interpolating gathers beside them are HBM-bound and listed in `hbm_ops`
            "upconv_bank_gemms": family(["pw_gemm", "pw_wgrad"]),
        }
        clocks, clock_src = inkernel_clocks()
        for name, ghz in (clocks or {}).items():         # secondary: against the peak at the clock the chip holds under that kernel
            if fam[name]["tflops"]:
                fam[name]["inkernel_clock_ghz"] = ghz
                fam[name]["frac_at_clock"] = fam[name]["tflops"] / (PEAK_BF16_TFLOPS * ghz / 2.4)
        conv_ms = sum(f["ms_per_step"] for f in fam.values())
        conv_gf = sum(f["gflop_per_step"] for f in fam.values())
        total_ms = sum(v[1] for v in summ.values()) * per_step
        for f in fam.values():
            f["share_of_step"] = f["ms_per_step"] / total_ms if total_ms > 0 else None
        fam_kernels = {
            "fwd_dgrad": "conv3x3_fwd_wres_kernel / conv3x3_fwd_ws_kernel / conv3x3_map4_kernel / conv3x3_map1_kernel "
                         "(conv3x3_fwd_kernel + splitk_finalize_kernel for other shapes) with the plain bias / ReLU epilogue",
            "fused_ln_fwd": "conv3x3_fwd_wres_kernel<.,2> / conv3x3_fwd_ws_kernel<.,2> (Conv2D -> LayerNorm -> ReLU in one "
                            "launch) and conv3x3_c3_fwd_kernel (first layer)",
            "wgrad": "conv3x3_wgrad_ws_kernel / conv3x3_wgrad_kernel + wgrad_reduce_kernel, conv3x3_c3_wgrad_kernel",
            "dgrad_relu_fused": "conv3x3_fwd_wres_kernel<.,3> (dgrad + ReLU-grad of the up-conv)",
            "dgrad_ln_bwd_fused": "conv3x3_fwd_wres_kernel<.,4> (dgrad + LayerNorm / ReLU backward of the layer below)",
            "upconv_bank_gemms": "pw_gemm_kernel, pw_wgrad_kernel + pw_wgrad_reduce_kernel (1x1 bank of the factored up-conv)",
        }
        dom_name = max(fam, key=lambda k: fam[k]["ms_per_step"])
        dom = fam[dom_name]
        dom_ops = {"fwd_dgrad": ["conv3x3_fwd"], "fused_ln_fwd": ["conv3x3_ln_relu_fwd", "conv3x3_c3_ln_relu_fwd"],
                   "wgrad": ["conv3x3_wgrad", "conv3x3_c3_wgrad"], "dgrad_relu_fused": ["conv3x3_dgrad_relu"],
                   "dgrad_ln_bwd_fused": ["conv3x3_dgrad_ln_bwd"], "upconv_bank_gemms": ["pw_gemm", "pw_wgrad"]}[dom_name]
        n_launch = sum(summ.get(k, (0, 0.0))[0] for k in dom_ops)
        ms = sum(summ.get(k, (0, 0.0))[1] for k in dom_ops)
        dom_bytes = sum(timer.nbytes(k) for k in dom_ops)
        # HBM-bound ops of the step, each with the bytes it must move (operands once) and the rate it moves them at
        hbm_ops = {}
        for k in ("upconv_gather_fwd", "upconv_gather_bwd", "resample", "resample_ln_bwd", "head_fwd", "head_ln_bwd", "adam_step"):
            if k in summ and timer.nbytes(k) > 0:
                cnt, t_ms = summ[k]
                hbm_ops[k] = {"launches_per_step": cnt * per_step, "ms_per_step": t_ms * per_step,
                              "algorithmic_gb_per_step": timer.nbytes(k) * per_step / 1e9,
                              "tb_per_s": timer.nbytes(k) / t_ms / 1e9, "frac_of_8_tb_per_s": timer.nbytes(k) / t_ms / 1e9 / 8.0}
        if args.breakdown:
            print(f"{'op family':<26}{'launches/step':>14}{'ms/step':>10}{'share':>8}", file=sys.stderr)
            for k, (cnt, t_ms) in sorted(summ.items(), key=lambda kv: -kv[1][1]):
                print(f"{k:<26}{cnt * per_step:>14.1f}{t_ms * per_step:>10.3f}{t_ms * per_step / total_ms:>8.1%}", file=sys.stderr)
            print(f"{'(sum of op events)':<26}{'':>14}{total_ms:>10.3f}", file=sys.stderr)
        img_s = batch * world * args.steps / elapsed
        ms_step = elapsed * 1000.0 / args.steps
        pmc = pmc_record(args.workload, args.dtype, batch)
        pmc_fam = {"fwd_dgrad": "conv3x3_fwd", "fused_ln_fwd": "conv3x3_ln_relu_fwd", "wgrad": "conv3x3_wgrad",
                   "dgrad_relu_fused": "conv3x3_dgrad_relu", "dgrad_ln_bwd_fused": "conv3x3_dgrad_ln_bwd",
                   "upconv_bank_gemms": "upconv_bank_gemms"}[dom_name]
        traffic = pmc["families"][pmc_fam]["hbm_bytes_per_launch"] if pmc and pmc_fam in pmc.get("families", {}) else None
        hbm_gb = pmc.get("hbm_bytes_per_step", 0.0) / 1e9 if pmc and pmc.get("hbm_bytes_per_step") else None
        line = {
            "metric": METRIC, "value": img_s, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: SR U-Net scale {scale} depth {depth} patch {patch} train step",
                       "global_batch": batch * world, "per_gpu_batch": batch, "params": model.count_params(),
                       "parallelism": f"dp{world}", "launch": "hipGraph replay" if graphed else "eager",
                       "conv_gflop_per_image_step": f_step / 1e9, "model_tflops": img_s * f_step / world / 1e12,
                       "final_loss": float(last_loss), "final_psnr": float(last_psnr)},
            "roofline": {"bound": "mfma",
                         # the conv kernel family with the largest share of the step (`families` has them all, each with its
                         # share); FLOPs as launched minus zero-padded channels, time = HIP events on the launch stream
                         "family": dom_name, "kernel": fam_kernels[dom_name],
                         "achieved": dom["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac"],
                         # HBM bytes per launch (stamped PMC passes) and the operand bytes per launch, both over the SAME launches:
                         # every launch of `traffic_ops` in a step (tools/pmc_summary.py books the same kernels to the family)
                         # which roof binds this family by the roofline model itself: arithmetic intensity (FLOPs per operand byte)
                         # against the ridge peak_flops / peak_bytes = 312.5 FLOP/B.  `frac` above stays the MFMA fraction the
                         # verdicts have tracked since r01; for a family below the ridge the binding ceiling is 8 TB/s x AI
                         "binding_roof": {"flop_per_byte": dom["flop_per_byte"], "ridge_flop_per_byte": PEAK_BF16_TFLOPS * 1e12 / 8e12,
                                          "bound": "hbm" if (dom["flop_per_byte"] or 1e9) < PEAK_BF16_TFLOPS * 1e12 / 8e12 else "mfma",
                                          "hbm_tb_per_s": dom["hbm_tb_per_s"], "frac_of_8_tb_per_s": dom["hbm_frac"]},
                         "traffic": traffic, "traffic_ops": dom_ops,
                         "algorithmic_bytes_per_launch": dom_bytes / n_launch if n_launch else None,
                         "launches_per_step": n_launch * per_step, "avg_launch_ms": ms / n_launch if n_launch else None,
                         "gflop_per_launch": dom["gflop_per_step"] / (n_launch * per_step) if n_launch else None,
                         "share_of_step": dom["share_of_step"],
                         # SURVEY 8d's step-level figure: images/s x the REFERENCE GRAPH's conv FLOPs per image-step / peak
                         # (every HBM-bound op included; the factored up-convs execute fewer FLOPs than the graph's 3x3
                         # convolutions they replace: `executed_gflop_per_step` is what the MFMA launches really did)
                         "frac_step": img_s / world * f_step / 1e12 / PEAK_BF16_TFLOPS,
                         # the same with the FLOPs the MFMA launches really execute (the factored up-convs do 1 / ratio^2 of
                         # the 3x3 convolutions they stand for): what the matrix pipes deliver over the whole step
                         "frac_step_executed": conv_gf / ms_step / PEAK_BF16_TFLOPS,      # GFLOP per ms = TFLOP/s
                         "algorithmic_gflop_per_step": f_step * batch / 1e9,
                         "executed_gflop_per_step": conv_gf,
                         "frac_all_conv_kernels": conv_gf / conv_ms / PEAK_BF16_TFLOPS if conv_ms > 0 else None,
                         # HBM side of the same step (stamped PMC passes, profiles/): GB per step through the fabric and
                         # that divided by the step time and 8 TB/s
                         "hbm_gb_per_step": hbm_gb,
                         "hbm_frac": hbm_gb / (ms_step / 1e3) / 8000.0 if hbm_gb else None,
                         "families": fam,
                         "inkernel_clock_source": clock_src,
                         "hbm_ops": hbm_ops,
                         "non_conv_ms_per_step": total_ms - conv_ms,
                         "timing": ("HIP events around every launch of %d eager steps run right after the graph-replayed "
                                    "timed region" % timed_steps) if graphed else "HIP events inside the timed region"},
        }
        if use_dist:
            line["ranks"] = dist.get_world_size()
            line["dist_backend"] = args.backend
            if args.backend == "nccl":
                line["rccl_ranks"] = dist.get_world_size()
            else:
                line["note"] = ("gloo rehearsal: %d ranks on %d GPU(s), gradients exchanged through host memory; value is NOT "
                                "a scaling measurement" % (world, torch.cuda.device_count()))
            line["exposed_comm_ms_per_step"] = exposed_ms
            line["exchange"] = "ad_allreduce_bucket" if model._dp._native is not None else "torch.distributed.all_reduce"
        if feed is not None:
            line["feed"] = feed
        if world == 1 and args.dtype == "bf16" and not args.no_micro:
            line["micro"] = micro_kernel(device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scale, depth, patch, args.workload)
        print(json.dumps(line), file=record_out, flush=True)
    if use_dist:
        try:
            model._dp.close()
            del step_fn, model            # graphs and side streams go before the communicator they reference
            torch.cuda.synchronize()
            dist.barrier()
        finally:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
