"""Shader clock and socket power while one kernel family runs back to back (rocm-smi, read-only), against the idle values.

    python tools/clock_probe.py            # 64 x 256^2, 64 -> 64, bf16: plain conv, fused LayerNorm conv, wgrad, a float4 copy

A worker thread keeps the GPU busy with one launch type for ~3 s; the main thread samples `rocm-smi --showclocks --showpower`
twice a second.  Says what clock the dense MFMA kernels actually run at (the peak in MI355X_MICROARCH.md is quoted at 2.4 GHz)."""
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adunet_amd import ops  # noqa: E402


def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
    mclk = re.search(r"mclk clock level:.*?\((\d+)Mhz\)", out)
    pw = re.search(r"Power \(W\):\s*([\d.]+)", out)
    return (int(sclk.group(1)) if sclk else None, int(mclk.group(1)) if mclk else None, float(pw.group(1)) if pw else None)


def main():
    dev = torch.device("cuda:0")
    n, hw, c = 64, 256, 64
    x = torch.randn((n, hw, hw, c), device=dev).bfloat16()
    dz = torch.randn((n, hw, hw, c), device=dev).bfloat16()
    w = torch.randn((3, 3, c, c), device=dev) * 0.05
    wf, _ = ops.conv3x3_pack(w, c, torch.bfloat16, want_dgrad=False)
    b = torch.zeros(c, device=dev)
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    dw = torch.zeros((3, 3, c, c), device=dev)
    ws = ops.Workspace(dev)
    big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
    big2 = torch.empty_like(big)
    jobs = [("idle", None),
            ("conv3x3 64->64 (plain epilogue)", lambda: ops.conv3x3_fwd(x, None, wf, b, c)),
            ("conv3x3 + LayerNorm + ReLU", lambda: ops.conv3x3_ln_relu_fwd(x, None, wf, b, gamma, beta, c, 1e-3)),
            ("wgrad 64->64", lambda: ops.conv3x3_wgrad(x, None, dz, dw, c, ws)),
            ("1 GiB copy (torch)", lambda: big2.copy_(big))]
    if "--step" in sys.argv:            # the whole K2' train step, replayed from its hipGraph
        import numpy as np
        from bench import WORKLOADS, synth_batch
        from adunet_amd.model import Adam, build_losses_and_metrics, build_super_resolution_unet
        scale, depth, patch, batch = WORKLOADS["K2p"]
        model, _ = build_super_resolution_unet(scale, depth_override=depth, input_size=patch, dtype=torch.bfloat16, device=dev)
        loss, metrics = build_losses_and_metrics("charbonnier")
        model.compile(optimizer=Adam(learning_rate=1e-4), loss=loss, metrics=metrics, jit_compile=False)
        model._require_device()
        model.set_weights(model.initial_weights(np.random.default_rng(1234), head_uniform=0.05))
        lr_img, hr_img = synth_batch(0, batch, patch, dev)
        step = model.make_graphed_train_step(lr_img, hr_img)
        jobs.append(("K2' train step (graph replay)", lambda: step(lr_img, hr_img)))
    print(f"{'job':<36}{'sclk MHz':>10}{'mclk MHz':>10}{'power W':>9}{'ms/launch':>11}")
    for name, fn in jobs:
        stop = threading.Event()
        count = [0]

        def work():
            while not stop.is_set():
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                count[0] += 20
        th = None
        if fn is not None:
            th = threading.Thread(target=work)
            th.start()
        time.sleep(1.0)
        c0, t0 = count[0], time.perf_counter()
        samples = []
        for _ in range(5):
            samples.append(smi())
            time.sleep(0.4)
        c1, t1 = count[0], time.perf_counter()
        if th is not None:
            stop.set()
            th.join()
        ok = [s for s in samples if s[0] is not None]
        sclk = sum(s[0] for s in ok) / max(len(ok), 1)
        mclk = sum(s[1] or 0 for s in ok) / max(len(ok), 1)
        pw = [s[2] for s in samples if s[2] is not None]
        ms = (t1 - t0) * 1e3 / max(c1 - c0, 1) if fn is not None else 0.0
        print(f"{name:<36}{sclk:>10.0f}{mclk:>10.0f}{(sum(pw) / len(pw) if pw else float('nan')):>9.0f}{ms:>11.3f}")
    print(subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout.strip()[-300:])


if __name__ == "__main__":
    main()
