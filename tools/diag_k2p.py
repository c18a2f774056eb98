#!/usr/bin/env python3
"""Diagnostic: per-tensor gradient errors of the K2' model (batch 2) against the oracle, fp32 and bf16-storage."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_model_gpu as T  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    which = sys.argv[1] if len(sys.argv) > 1 else "f32"
    dtype = torch.float32 if which == "f32" else torch.bfloat16
    scale, depth, p, n = (0.25, 4, 256, 2) if len(sys.argv) < 3 else (float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    oracle, params, model, rng = T.build_pair(scale, depth, p, dtype, dev)
    lr, hr = T.synth(rng, n, p)
    want_loss, want_grads, want_out, want_psnr = oracle.loss_and_grads(
        params, lr.astype(np.float64), hr.astype(np.float64), storage=T.storage_of(model, n))
    out, loss, psnr, (tape, x, t) = model.forward_loss(lr, hr, keep=True)
    model._backward(tape, x, t, 1.0 / x.numel())
    print("out rel", T.rel(out.cpu().numpy(), want_out), "loss", float(loss), want_loss, "psnr", float(psnr), want_psnr)
    grads = model.get_grads()
    rows = sorted(((T.rel(grads[k], want_grads[k]), k) for k in want_grads), reverse=True)
    for r, k in rows[:12]:
        e = np.abs(grads[k].astype(np.float64) - want_grads[k])
        flat = np.sort(e.reshape(-1))[::-1]
        print(f"{k:<32}{r:10.3e}  max|want| {np.abs(want_grads[k]).max():.3e}  err top/100th/median "
              f"{flat[0]:.2e} {flat[min(99, flat.size - 1)]:.2e} {np.median(flat):.2e}  argmax {np.unravel_index(e.argmax(), e.shape)}")
    print("...")
    for r, k in rows[-3:]:
        print(f"{k:<32}{r:10.3e}")


if __name__ == "__main__":
    main()
