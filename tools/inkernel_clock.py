#!/usr/bin/env python3
"""In-kernel shader clock of the dense conv launches (MI355X_MICROARCH.md "DVFS give-back" item 6).

    tools/build_variant.sh clock -DAD_CLOCK            (in the container)
    ADUNET_LIB=ab/clock.so python tools/inkernel_clock.py [--seconds 2.5] [--json out.json]      (on the GPU box)

For each kernel type: >= `seconds` of back-to-back launches on random data (the clock the firmware settles at under THAT load),
then the (s_memtime cycles, s_memrealtime 10-ns ticks) pairs that MFMA wave 0 of every workgroup of the LAST launches left in
the diagnostic buffer: clock = cycles / wall, median over the workgroups.  Beside it the HIP-event time per launch of the same
loop and the resulting fraction of (a) the 2.5 PFLOP/s dense bf16 peak at 2.4 GHz (what VERDICT grades against) and (b) that
peak scaled to the measured clock (what the matrix pipes could deliver at the clock the chip allows this launch).
The stamped build is a diagnostic: its times are not quoted as performance (the shipped library has no stamps).
"""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from adunet_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=2.5)
ap.add_argument("--json", default=None)
ap.add_argument("--n", type=int, default=64)
args = ap.parse_args()
lib = _lib.load()
try:
    rd = lib.ad_dbg_clock_read
except AttributeError:
    sys.exit("this library was not built with -DAD_CLOCK (tools/build_variant.sh clock -DAD_CLOCK; ADUNET_LIB=ab/clock.so)")
rd.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda:0")
n, hw, c = args.n, 256, 64
g = torch.Generator(device="cpu").manual_seed(3)
rnd = lambda *shape: (torch.rand(shape, generator=g) * 2 - 1)
x = rnd(n, hw, hw, c).to(dev).bfloat16()
dz = rnd(n, hw, hw, c).to(dev).bfloat16()
z = rnd(n, hw, hw, c).to(dev).bfloat16()
w = (rnd(3, 3, c, c) * 0.05).to(dev)
wf, wd = ops.conv3x3_pack(w, c, torch.bfloat16)
b = torch.zeros(c, device=dev)
gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
mean = torch.zeros(n * hw * hw, device=dev)
rstd = torch.ones(n * hw * hw, device=dev)
dgam, dbet, dbias = (torch.empty(c, device=dev) for _ in range(3))
dw = torch.empty_like(w)
ws = ops.Workspace(dev)
flops = 2.0 * n * hw * hw * 9 * c * c
jobs = [
    ("conv3x3_fwd_wres_kernel<.,0>  plain 64->64", lambda: ops.conv3x3_fwd(x, None, wf, b, c)),
    ("conv3x3_fwd_wres_kernel<.,2>  conv + LayerNorm + ReLU", lambda: ops.conv3x3_ln_relu_fwd(x, None, wf, b, gamma, beta, c, 1e-3)),
    ("conv3x3_fwd_wres_kernel<.,4>  dgrad + LayerNorm backward", lambda: ops.conv3x3_dgrad_ln_bwd(dz, wd, z, mean, rstd, gamma, beta, dgam, dbet, dbias, ws)),
    ("conv3x3_wgrad_ws_kernel       wgrad 64->64", lambda: ops.conv3x3_wgrad(x, None, dz, dw, c, ws)),
]
buf = (ctypes.c_ulonglong * (2 * 4096))()
out = {"shape": f"{n} x {hw} x {hw}, {c} -> {c}, bf16, random data", "seconds_of_load_before_the_stamp": args.seconds, "kernels": {}}
print(f"{'kernel':<58}{'ms/launch':>10}{'clock GHz':>10}{'TFLOP/s':>9}{'of 2.5 PF':>10}{'of peak at clock':>17}")
for name, fn in jobs:
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    launches = 0
    while time.time() - t0 < args.seconds:
        for _ in range(50):
            fn()
        launches += 50
        torch.cuda.synchronize()
    rd(buf, 4096, 1)                                  # drop what the load phase left
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    rd(buf, 4096, 0)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 2).astype(np.float64)
    a = a[a[:, 1] > 0]
    ghz = a[:, 0] / (a[:, 1] * 10.0)                  # cycles per ns
    clock = float(np.median(ghz))
    tf = flops / ms / 1e9
    key = name.split()[0]
    out["kernels"][key] = {
        "label": name, "ms_per_launch": ms, "clock_ghz_median": clock, "clock_ghz_p10_p90": [float(np.percentile(ghz, 10)), float(np.percentile(ghz, 90))],
        "workgroups": int(len(a)), "tflops": tf, "frac_of_2p5_pf": tf / 2500.0, "frac_of_peak_at_clock": tf / (2500.0 * clock / 2.4),
        "launches_before_stamp": launches}
    print(f"{name:<58}{ms:>10.4f}{clock:>10.3f}{tf:>9.0f}{tf / 2500.0:>10.3f}{tf / (2500.0 * clock / 2.4):>17.3f}")
if args.json:
    json.dump(out, open(args.json, "w"), indent=1)
