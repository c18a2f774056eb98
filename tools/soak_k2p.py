import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from adunet_amd.model import Adam, build_losses_and_metrics, build_super_resolution_unet
dev = torch.device("cuda:0")
for dtype in (torch.bfloat16, torch.float16):
    rng = np.random.default_rng(0)
    model, info = build_super_resolution_unet(0.25, depth_override=4, input_size=256, dtype=dtype, device=dev)
    loss, metrics = build_losses_and_metrics("charbonnier")
    model.compile(optimizer=Adam(1e-4), loss=loss, metrics=metrics)
    model._require_device()
    model.set_weights(model.initial_weights(np.random.default_rng(1), head_uniform=0.05))
    batch = 64
    hr = rng.random((batch, 256, 256, 3), dtype=np.float32)
    hr = (hr + np.roll(hr, 1, 1) + np.roll(hr, 1, 2) + np.roll(hr, 2, 1)) / 4.0      # some structure
    lr = np.clip(hr + 0.08 * rng.standard_normal(hr.shape, dtype=np.float32), 0, 1).astype(np.float32)
    step = model.make_graphed_train_step(lr, hr)
    x, y = torch.from_numpy(lr).to(dev), torch.from_numpy(hr).to(dev)
    t0 = time.perf_counter(); out = []
    for i in range(int(os.environ.get("SOAK_STEPS", "400"))):
        r = step(x, y)
        if i % 50 == 0:
            out.append((i, float(r[0]), float(r[1])))
    torch.cuda.synchronize()
    print(dtype, f"{(time.perf_counter()-t0)/max(i+1,1)*1e3:.2f} ms/step", out[:3], "...", out[-2:], flush=True)
    assert all(np.isfinite(v[1]) for v in out) and out[-1][1] < out[0][1]
    del model, step; torch.cuda.empty_cache()
