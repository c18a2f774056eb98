#!/usr/bin/env python3
"""SR inference rate (model(x): forward only, evaluate_model.py's loop body), K2' shape, bf16 -- eager launches and one
hipGraph replay; ADUNET_INFER_KEEP_Z=1 restores the forward pass that also stores z and the LayerNorm statistics (A/B)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adunet_amd.model import build_super_resolution_unet

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model, _ = build_super_resolution_unet(0.25, depth_override=4, input_size=256, dtype=torch.bfloat16, device=dev)
model._require_device()
model.set_weights(model.initial_weights(rng, head_uniform=0.05))
x = torch.from_numpy(rng.random((batch, 256, 256, 3), dtype=np.float32)).to(dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for keep in ("1", "0", "1", "0"):
    os.environ["ADUNET_INFER_KEEP_Z"] = keep
    ms = timeit(lambda: model(x))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        model(x)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model(x)
    mg = timeit(g.replay)
    print(f"KEEP_Z={keep}: eager {ms:6.3f} ms {batch / ms * 1e3:8.0f} img/s | graph replay {mg:6.3f} ms {batch / mg * 1e3:8.0f} img/s")
    del g, out
