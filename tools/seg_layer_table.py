"""Per-launch table of one eager train step of a segmentation workload (bench.py's K3 / K3d4 / K3ln): every ops call with
its tensor shapes and HIP-event time; conv rows also carry TFLOP/s and the fraction of the bf16 peak.

    python tools/seg_layer_table.py --workload K3d4"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from adunet_amd import ops, seg_model as S  # noqa: E402
from bench import PEAK_BF16_TFLOPS, SEG_WORKLOADS  # noqa: E402

rows = []
NAMES = ["conv3x3_fwd", "conv3x3_wgrad", "conv3x3_ln_relu_fwd", "batchnorm_relu_fwd_train", "batchnorm_relu_pool_fwd_train",
         "batchnorm_relu_bwd", "layernorm_relu_bwd", "maxpool2_fwd", "maxpool2_bwd", "resample", "seg_head_fwd", "seg_head_bwd",
         "pad_channels", "conv_transpose2x2s2_fwd", "conv_transpose2x2s2_bwd"]


def wrap(name):
    fn = getattr(ops, name, None)
    if fn is None:
        return

    def inner(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kw)
        e1.record()
        shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor) and a.dim() == 4]
        rows.append((name, shapes, kw.get("cout", None), args, e0, e1))
        return out

    setattr(ops, name, inner)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="K3d4")
    a = ap.parse_args()
    norm, depth, base, batch, proto_name = SEG_WORKLOADS[a.workload]
    dev = torch.device("cuda:0")
    m = (S.build_adaptive_depth_unet(256, base, depth, dtype=torch.bfloat16, device=dev) if norm == "bn"
         else S.build_unet(256, 1, base, depth, dtype=torch.bfloat16, device=dev))
    proto = S.PROTOCOLS[proto_name]
    m.compile(optimizer=S.build_optimizer(proto, steps_per_epoch=100, epochs=2), loss=proto.loss_builder())
    m._require_device()
    rng = np.random.default_rng(0)
    img = rng.random((batch, 256, 256, 3), dtype=np.float32)
    mask = (rng.random((batch, 256, 256, 1)) < 0.35).astype(np.float32)
    for n in NAMES:
        wrap(n)
    for _ in range(3):
        rows.clear()
        m.train_on_batch(img, mask)
        torch.cuda.synchronize()
    tot = 0.0
    print(f"{'op':<32}{'shapes':<60}{'ms':>8}{'TFLOP/s':>9}{'frac':>7}")
    for name, shapes, cout, args, e0, e1 in rows:
        ms = e0.elapsed_time(e1)
        tot += ms
        extra = ""
        if name in ("conv3x3_fwd", "conv3x3_ln_relu_fwd") and shapes:
            n, h, w, _ = shapes[0]
            cin = sum(s[-1] for s in shapes)
            co = next((x for x in args if isinstance(x, int)), None)
            if co:
                tf = 2.0 * n * h * w * cin * co * 9 / ms / 1e9
                extra = f"{tf:>9.0f}{tf / PEAK_BF16_TFLOPS:>7.3f}  cout={co}"
        if name == "conv3x3_wgrad" and len(shapes) >= 2:
            n, h, w, _ = shapes[0]
            cin = sum(s[-1] for s in shapes[:-1])
            co = shapes[-1][-1]
            tf = 2.0 * n * h * w * cin * co * 9 / ms / 1e9
            extra = f"{tf:>9.0f}{tf / PEAK_BF16_TFLOPS:>7.3f}"
        print(f"{name:<32}{str(shapes):<60}{ms:>8.3f}{extra}")
    print(f"(sum) {tot:.3f} ms")


if __name__ == "__main__":
    main()
