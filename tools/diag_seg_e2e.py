#!/usr/bin/env python3
"""Diagnostic: per-tensor end-to-end gradient errors of a segmentation model against the oracle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_seg_model_gpu as T  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    kind, which, p, depth, batch = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    dtype = torch.float32 if which == "f32" else torch.bfloat16
    S, model, oracle, params, state, img, mask = T.build(kind, dtype, dev, p=p, depth=depth, batch=batch)
    proto = S.PROTOCOLS["A"]
    model.compile(optimizer=S.build_optimizer(proto, steps_per_epoch=10, epochs=2), loss=proto.loss_builder())
    st = dict(state)
    want_loss, grads, pw, dice, iou = oracle.loss_and_grads(params, st, img.astype(np.float64), mask.astype(np.float64), 0.4, 0.6,
                                                            storage=T.storage_of(model, batch))
    x, m = model._to_dev(img), model._to_dev_mask(mask)
    prob, sums, tape = model._forward_seg(x, m, training=True, keep=True)
    model._backward_seg(tape, m)
    pe = np.abs(prob.cpu().numpy() - pw)
    print("prob max err", pe.max(), "mean err", pe.mean(), "loss", float(model._metrics_from(sums, float(m.numel()))[0]), want_loss)
    got = model.get_grads()
    rows = sorted(((T.rel(got[k], grads[k]), k) for k in grads if np.abs(grads[k]).max() > 1e-9), reverse=True)
    for r, k in rows[:14]:
        e = np.abs(got[k].astype(np.float64) - grads[k])
        flat = np.sort(e.reshape(-1))[::-1]
        print(f"{k:<34}{r:10.3e}  max|want| {np.abs(grads[k]).max():.3e}  err top/100th/median {flat[0]:.2e} "
              f"{flat[min(99, flat.size - 1)]:.2e} {np.median(flat):.2e}")
    ga = np.concatenate([got[k].reshape(-1) for k in grads]).astype(np.float64)
    gb = np.concatenate([grads[k].reshape(-1) for k in grads])
    print("cosine", float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb))))
    # near-kink census in the oracle's forward
    tot = 0
    for rec in oracle._tape:
        if rec[0] == "cna":
            _, c, nn, xin, (zs, mu, rstd), a = rec
            shp = (1, 1, 1, -1) if np.ndim(mu) == 1 else mu.shape
            y = (zs - np.reshape(mu, shp)) * np.reshape(rstd, shp) * params[nn + "/gamma"] + params[nn + "/beta"]
            k = int((np.abs(y) < 1e-6).sum())
            tot += k
            if k:
                print(f"  {c}: {k} pre-activations within 1e-6 of the ReLU kink ({y.size} elements)")
    print("near-kink elements:", tot)


if __name__ == "__main__":
    main()
