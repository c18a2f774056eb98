#!/usr/bin/env python3
"""Diagnostic: layer-by-layer forward comparison of a segmentation model against the oracle (bf16: storage mode)."""
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
    storage = T.storage_of(model, batch)
    st = dict(state)
    pw = oracle.forward(params, st, img.astype(np.float64), training=True, storage=storage)
    x, m = model._to_dev(img), model._to_dev_mask(mask)
    prob, sums, tape = model._forward_seg(x, m, training=True, keep=True)
    print("prob rel", T.rel(prob.cpu().numpy(), pw))
    otape = [r for r in oracle._tape if r[0] == "cna"]
    ptape = [r for r in tape if r[0] == "cna"]
    for (_, c, nn, xin, (zs, mu, rstd), a), (_, cs, nn2, x1, x2, z, mean, rstd_p) in zip(otape, ptape):
        zz = z.float().cpu().numpy()
        xin_p = x1.float().cpu().numpy()[..., :xin.shape[-1]] if x2 is None else np.concatenate(
            [x1.float().cpu().numpy(), x2.float().cpu().numpy()], axis=-1)
        e_in = T.rel(xin_p, xin)
        e_z = T.rel(zz, zs)
        nflip = int((np.abs(zz - zs) > 1e-6 * (np.abs(zs) + 1e-3)).sum())
        e_mu = T.rel(mean.cpu().numpy().reshape(-1), np.asarray(mu).reshape(-1))
        e_rs = T.rel(rstd_p.cpu().numpy().reshape(-1), np.asarray(rstd).reshape(-1))
        print(f"{c:<12} in {e_in:9.2e}  z {e_z:9.2e} (differing {nflip}/{zz.size})  mean {e_mu:9.2e}  rstd {e_rs:9.2e}  shape {zz.shape}")


if __name__ == "__main__":
    main()
