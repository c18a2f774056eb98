"""N>1 path on CPU: world_size-2 `gloo` runs of the data-parallel gradient exchange (adunet_amd.parallel).

The HIP model itself needs a GPU, so the exchange is exercised on a stand-in that exposes the same flat
buffers (P, G, index) -- the DataParallel class touches nothing else -- and the mathematical contract
(2 ranks x batch b == 1 rank x batch 2b, LayerNorm has no cross-sample statistic) is checked with the oracle's
gradients flowing through the real bucketed all-reduce.
"""
import os
import socket
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.sr_unet import SRUNetOracle


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FlatStandIn:
    """Same flat-buffer surface as adunet_amd.model.Model, on CPU tensors."""

    def __init__(self, shapes):
        self.index = OrderedDict()
        off = 0
        for name, shape in shapes.items():
            self.index[name] = (off, tuple(shape))
            off += int(np.prod(shape))
        self.P = torch.zeros(off)
        self.G = torch.zeros(off)
        self.grad_ready = self.grad_sync = None

    def count_params(self):
        return self.P.numel()

    def put(self, buf, values):
        for name, (off, shape) in self.index.items():
            buf[off:off + int(np.prod(shape))] = torch.tensor(values[name], dtype=torch.float32).reshape(-1)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adunet_amd.parallel import DataParallel
    torch.set_num_threads(2)
    oracle = SRUNetOracle(0.5, 1, 16, base_channels=8, residual_head_channels=8)
    rng = np.random.default_rng(7)
    params = oracle.init_params(rng, head_uniform=0.05)
    hr = rng.random((4, 16, 16, 3))
    lr = np.clip(hr + 0.05 * rng.standard_normal(hr.shape), 0, 1)
    model = FlatStandIn(oracle.param_shapes)
    if rank == 0:
        model.put(model.P, params)                       # rank 0 owns the initial weights
    dp = DataParallel(model, bucket_bytes=4096)          # small buckets => several collectives
    assert len(dp.buckets) > 2
    got_p = {n: model.P[o:o + int(np.prod(s))].numpy().reshape(s) for n, (o, s) in model.index.items()}
    bcast_ok = all(np.allclose(got_p[k], params[k], atol=1e-6) for k in params)
    # each rank: its half of the global batch, loss gradient scaled by 1/(local elements)
    sl = slice(rank * 2, rank * 2 + 2)
    _, grads, _, _ = oracle.loss_and_grads(params, lr[sl], hr[sl])
    model.put(model.G, grads)
    for name in reversed(list(model.index)):             # backward finishes gradients from the end of the buffer
        model.grad_ready(model.index[name][0])
    gscale = model.grad_sync(model)
    _, full, _, _ = oracle.loss_and_grads(params, lr, hr)   # one rank, global batch
    err = 0.0
    for name, (off, shape) in model.index.items():
        g = model.G[off:off + int(np.prod(shape))].numpy().reshape(shape) * gscale
        err = max(err, float(np.abs(g - full[name]).max() / (np.abs(full[name]).max() + 1e-30)))
    # second step reuses the same buckets (state reset)
    model.put(model.G, grads)
    model.grad_ready(0)
    g2 = model.grad_sync(model)
    q.put((rank, bcast_ok, gscale, err, g2, dp._next, len(dp._works)))
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_equals_global_batch():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, bcast_ok, gscale, err, g2, nxt, nworks in res:
        assert bcast_ok, f"rank {rank}: weights were not broadcast from rank 0"
        assert gscale == 0.5 and g2 == 0.5
        assert err < 1e-6, f"rank {rank}: averaged gradient differs from the global-batch gradient ({err})"
        assert nxt == 0 and nworks == 0


def _bank_worker(rank, world, port, q):
    """BASELINE config 5 under data parallelism (adunet_amd.multitask.AdaptiveDepthBank.data_parallel): ONE exchange object per
    model of the bank, attached when the model is built -- before or after data_parallel() was called."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from adunet_amd import multitask as M

    class StandInBank(M.AdaptiveDepthBank):           # the HIP models need a GPU: same flat-buffer surface on CPU tensors
        def _build_sr(self, key):
            m = FlatStandIn({"conv/kernel": (3, 3, 8, 8 * key[1]), "conv/bias": (8 * key[1],), "head/kernel": (8, 3)})
            m.P += 100.0 * rank + key[1]                # every rank starts elsewhere: the broadcast must bring rank 0's values
            return m

        def _build_seg(self, steps_per_epoch, epochs):
            m = FlatStandIn({"seg/kernel": (3, 3, 3, 16), "seg/bias": (16,)})
            m.P += 100.0 * rank + 77.0
            return m

    bank = StandInBank(input_size=256, dtype=torch.float32, device=None)
    _, early = bank.sr_model(0.5)                       # built BEFORE data_parallel(): attached by the call
    assert getattr(early, "_dp", None) is None
    bank.data_parallel(bucket_bytes=1024)
    key3, late = bank.sr_model(0.3)                     # built AFTER: attached on construction
    seg = bank.seg_model()
    _, again = bank.sr_model(0.5)
    ok = again is early and key3 == (0.3, 2) and len(bank.dps) == 3 and len({id(m._dp) for m in (early, late, seg)}) == 3
    ok &= all(m._dp.world == world and m.grad_ready is not None and m.grad_sync is not None for m in (early, late, seg))
    # rank 0's weights everywhere
    ok &= bool((early.P == 3.0).all()) and bool((late.P == 2.0).all()) and bool((seg.P == 77.0).all())
    # one model's exchange leaves the others' buffers alone, and sums over the ranks
    for i, m in enumerate((early, late, seg)):
        m.G += float((rank + 1) * (i + 1))
    late.grad_ready(0)
    gs = late.grad_sync(late)
    ok &= gs == 0.5 and bool((late.G == 3.0 * 2).all()) and bool((early.G == float(rank + 1)).all()) and bool((seg.G == 3.0 * (rank + 1)).all())
    q.put((rank, bool(ok)))
    bank.close()
    dist.destroy_process_group()


def test_bank_of_models_gets_one_exchange_object_each():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
