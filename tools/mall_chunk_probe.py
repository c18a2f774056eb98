#!/usr/bin/env python3
"""Does running consecutive full-resolution launches per IMAGE CHUNK (so that the tensor one launch writes is still in the
256 MiB Infinity Cache when the next launch reads it) beat running each launch over the whole batch?  (r04 probe; no kernel
changes: the same entry points on batch slices.)

    python tools/mall_chunk_probe.py            # K2' full-resolution shapes: 64 x 256 x 256 x 64, bf16

forward pair : conv+LN+ReLU (x -> z1, a1), conv+LN+ReLU (a1 -> z2, a2)
backward trio: dgrad+LN-backward (dz2, z1 -> dz1) ; wgrad (a0, dz1) ; dgrad+LN-backward (dz1, z0 -> dz0)   [dz1 written once, read twice]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adunet_amd import ops

dev = torch.device("cuda:0")
n, hw, c = 64, 256, 64
g = torch.Generator(device="cpu").manual_seed(3)
rnd = lambda *s: (torch.rand(s, generator=g) * 2 - 1)
x = rnd(n, hw, hw, c).to(dev).bfloat16()
w1, w2 = (rnd(3, 3, c, c) * 0.05).to(dev), (rnd(3, 3, c, c) * 0.05).to(dev)
wf1, wd1 = ops.conv3x3_pack(w1, c, torch.bfloat16)
wf2, wd2 = ops.conv3x3_pack(w2, c, torch.bfloat16)
b = torch.zeros(c, device=dev)
gam, bet = torch.ones(c, device=dev), torch.zeros(c, device=dev)
ws = ops.Workspace(dev)
dgam, dbet, dbias = (torch.empty(c, device=dev) for _ in range(3))
dw = torch.empty_like(w1)


def fwd_pair(xs):
    z1, a1, m1, r1 = ops.conv3x3_ln_relu_fwd(xs, None, wf1, b, gam, bet, c)
    z2, a2, m2, r2 = ops.conv3x3_ln_relu_fwd(a1, None, wf2, b, gam, bet, c)
    return z1, a1, m1, r1, z2


def bwd_trio(dz2, z1, m1, r1, a0, z0, m0, r0):
    dz1 = ops.conv3x3_dgrad_ln_bwd(dz2, wd2, z1, m1, r1, gam, bet, dgam, dbet, dbias, ws)
    ops.conv3x3_wgrad(a0, None, dz1, dw, c, ws)
    return ops.conv3x3_dgrad_ln_bwd(dz1, wd1, z0, m0, r0, gam, bet, dgam, dbet, dbias, ws)


def timeit(fn, iters=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


z1, a1, m1, r1, z2 = fwd_pair(x)
dz2 = rnd(n, hw, hw, c).to(dev).bfloat16()
z0, m0, r0 = z2, m1, r1          # any tensors of the right shape
print(f"{'images per chunk':>18}{'forward pair ms':>18}{'backward trio ms':>18}")
for chunk in (64, 32, 16, 8, 4):
    sl = [slice(i, i + chunk) for i in range(0, n, chunk)]
    px = lambda t, s: t[s.start * hw * hw:s.stop * hw * hw]
    f = timeit(lambda: [fwd_pair(x[s]) for s in sl])
    bt = timeit(lambda: [bwd_trio(dz2[s], z1[s], px(m1, s), px(r1, s), x[s], z0[s], px(m0, s), px(r0, s)) for s in sl])
    print(f"{chunk:>18}{f:>18.3f}{bt:>18.3f}")

# ---- a chain of FOUR conv + LayerNorm + ReLU layers (K2' forward: dec0a -> dec0b -> head_a -> head_b), replayed from a hipGraph so
# that launch gaps do not blur the comparison
print()
print(f"{'images per chunk':>18}{'4-layer forward chain, graph replay, ms':>44}")
packs = [ops.conv3x3_pack((rnd(3, 3, c, c) * 0.05).to(dev), c, torch.bfloat16)[0] for _ in range(4)]


def chain(xs):
    a = xs
    keep = []
    for wf in packs:
        z, a, m, r = ops.conv3x3_ln_relu_fwd(a, None, wf, b, gam, bet, c)
        keep.append((z, m, r))
    return a, keep


for chunk in (64, 32, 16):
    sl = [slice(i, i + chunk) for i in range(0, n, chunk)]
    run = lambda: [chain(x[s]) for s in sl]
    run(); torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        outs = run()
    for _ in range(5):
        gph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        gph.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{chunk:>18}{e0.elapsed_time(e1) / 20:>44.3f}")
    del gph, outs
