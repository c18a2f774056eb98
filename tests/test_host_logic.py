"""CPU tests of the host logic: depth heuristics (known answers from SURVEY 8 a7), layer configs, resize tap
tables, bucket planning, and that the C-ABI library loads and exports every symbol of include/adunet.h."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from adunet_amd import _lib
    header = open(os.path.join(ROOT, "include", "adunet.h")).read()
    declared = set(re.findall(r"\b(ad_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()           # no GPU needed to load; AttributeError if a symbol is missing
    assert lib.ad_version() >= 1
    assert lib.ad_cin_granule(_lib.AD_BF16) == 32 and lib.ad_cin_granule(_lib.AD_F32) == 16
    assert lib.ad_conv3x3_wgrad_ws_bytes(2, 16, 16, 64, 64, _lib.AD_BF16) > 0
    # explicit options instead of environment reads inside the library; shape queries of the r03 entry points
    assert lib.ad_get_option(b"no_map4") == 0 and lib.ad_set_option(b"no_map4", 1) == 0 and lib.ad_get_option(b"no_map4") == 1
    assert lib.ad_set_option(b"no_map4", 0) == 0 and lib.ad_set_option(b"bogus", 1) != 0 and lib.ad_get_option(b"bogus") == -1
    assert lib.ad_device_cus() % 64 == 0 and lib.ad_device_cus() >= 64
    assert lib.ad_pw_supported(4096, 128, 576, _lib.AD_BF16) == 1 and lib.ad_pw_supported(4096, 96, 576, _lib.AD_BF16) == 0
    assert lib.ad_pw_wgrad_supported(4096, 128, 64, _lib.AD_BF16) == 1 and lib.ad_pw_wgrad_supported(4096, 128, 64, _lib.AD_F32) == 0
    assert lib.ad_upconv_gather_fwd_supported(64, 7, _lib.AD_BF16) == 1 and lib.ad_upconv_gather_fwd_supported(1024, 4, _lib.AD_BF16) == 0
    assert lib.ad_upconv_gather_bwd_supported(8) == 1 and lib.ad_upconv_gather_bwd_supported(15) == 0
    # the forward gather's staging rule lives in the library: x4 table 64 -> 256 at 64 channels = 16 output columns
    # per workgroup + neighbours read 6 low-resolution columns; a table that goes backwards / out of range is refused
    sx = np.clip((np.arange(256) - 1.5) // 4, 0, 63).astype(np.int32)
    assert lib.ad_upconv_slab_cols(sx.ctypes.data, 64, 256, 64, _lib.AD_BF16) == 6
    assert lib.ad_upconv_slab_cols(sx[::-1].copy().ctypes.data, 64, 256, 64, _lib.AD_BF16) == -1
    assert lib.ad_upconv_slab_cols(sx.ctypes.data, 32, 256, 64, _lib.AD_BF16) == -1


def test_mosaic_planner_takes_the_mosaic_where_it_saves_a_round():
    """Host arithmetic of the image mosaic (conv.hip plan_mosaic; without a device the library plans for 256 CUs, as MI355X):
    the levels of Experiment 2's pyramids (run_experiment_adaptive_depth.sh:47-55: scale 0.6 at batch 32, 0.7 at batch 8)."""
    from adunet_amd import _lib
    lib = _lib.load()
    bf = _lib.AD_BF16
    # scale 0.6, batch 32: 34-wide (18 -> 11 rounds) and 56-wide (16 -> 14) levels; 93 keeps its 18 rounds
    assert lib.ad_conv3x3_mosaic(32, 34, 34, 1024, 0, 1024, bf, 0) > 0 and lib.ad_conv3x3_mosaic(32, 34, 34, 1024, 0, 1024, bf, 1) > 0
    assert lib.ad_conv3x3_mosaic(32, 56, 56, 512, 0, 512, bf, 0) > 0
    assert lib.ad_conv3x3_mosaic(32, 93, 93, 256, 0, 256, bf, 0) == 0
    # scale 0.7, batch 8: every level keeps its 8 or 9 rounds
    for hw, c in ((180, 128), (126, 256), (89, 512), (63, 1024), (45, 2048)):
        assert lib.ad_conv3x3_mosaic(8, hw, hw, c, 0, c, bf, 0) == 0, hw
    # whole tiles, one image, float32 (generic kernels), the option
    assert lib.ad_conv3x3_mosaic(64, 64, 64, 128, 0, 128, bf, 0) == 0 and lib.ad_conv3x3_mosaic(1, 34, 34, 1024, 0, 1024, bf, 0) == 0
    assert lib.ad_conv3x3_mosaic(32, 34, 34, 1024, 0, 1024, _lib.AD_F32, 0) == 0
    lib.ad_set_option(b"no_mosaic", 1)
    try:
        assert lib.ad_conv3x3_mosaic(32, 34, 34, 1024, 0, 1024, bf, 0) == 0
    finally:
        lib.ad_set_option(b"no_mosaic", 0)
    # the row length the planner picks tiles the mosaic in the fewest rounds: 32 images of 34 x 34 -> 4 x 8 (9 x 18 tiles)
    assert lib.ad_conv3x3_mosaic(32, 34, 34, 1024, 0, 1024, bf, 0) in (4, 8)


def test_missing_library_fails_loudly(monkeypatch):
    from adunet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libadunet_hip.so")
    with pytest.raises(_lib.AdunetError):
        _lib.load()


KNOWN_256 = {0.2: 2, 0.25: 2, 0.3: 3, 0.4: 3, 0.45: 4, 0.5: 4, 0.6: 6, 0.7: 7, 0.8: 7, 0.9: 7}


@pytest.mark.parametrize("mod", ["product", "oracle"])
def test_depth_heuristics_known_answers(mod):
    if mod == "product":
        from adunet_amd import custom_layers as m
    else:
        from oracle import ops as m
    for s, d in KNOWN_256.items():
        assert m.custom_depth_from_scale(s) == d, s
    assert m.custom_depth_from_scale(0.25, base_resolution=64) == 1
    assert m.custom_depth_from_scale(0.5, base_resolution=64) == 2
    assert m.custom_depth_from_scale(0.25, base_resolution=128) == 2
    assert m.custom_depth_from_scale(0.5, base_resolution=128) == 3
    assert m.custom_depth_from_scale(0.25, base_resolution=512) == 3
    assert m.custom_depth_from_scale(0.5, base_resolution=512) == 5
    assert [m.infer_depth_from_scale(s) for s in (0.1, 0.25, 0.3, 0.45, 0.5, 0.9)] == [1, 1, 2, 2, 3, 3]
    assert m.infer_depth_from_scale(0.9, max_depth=2) == 2
    assert m.estimate_bottleneck_size(256, 0.6, 4) == 33          # round, not ceil: differs from the real pyramid (34)
    assert m.depth_and_sizes(0.5)[1][:4] == [256, 128, 64, 32]
    for bad in (0.05, 1.0, -1, 0.0):
        with pytest.raises(ValueError):
            m.custom_depth_from_scale(bad)
        with pytest.raises(ValueError):
            m.infer_depth_from_scale(bad)
    with pytest.raises(ValueError):
        m.custom_depth_from_scale(0.5, min_depth=0)


def test_layer_configs_mirror_reference():
    from adunet_amd.custom_layers import ClipAdd, ClippedResidualAdd, ResizeByScale, ResizeToMatch, SERIALIZED_NAMES
    l = ResizeByScale(0.6, name="enc_down")
    cfg = l.get_config()
    assert cfg["scale"] == 0.6 and cfg["method"] == "bilinear" and cfg["antialias"] is True and cfg["name"] == "enc_down"
    assert ResizeToMatch(name="dec_up").get_config()["antialias"] is True
    assert ClipAdd is ClippedResidualAdd
    assert SERIALIZED_NAMES["ResizeByScale"] == "resize>ResizeByScale"
    # pyramid chains pinned by the reference summaries (float32 ceil)
    for scale, chain in [(0.6, [256, 154, 93, 56, 34]), (0.7, [256, 180, 126, 89, 63, 45]), (0.2, [256, 52, 11]),
                         (0.3, [256, 77, 24, 8]), (0.4, [256, 103, 42, 17]), (0.8, [256, 205, 164, 132]),
                         (0.9, [256, 231, 208, 188])]:
        lay = ResizeByScale(scale)
        h = 256
        for want in chain[1:]:
            h = lay.output_hw(h, h)[0]
            assert h == want, (scale, chain)


@pytest.mark.parametrize("sizes", [(256, 154), (154, 93), (256, 64), (64, 16), (16, 4), (4, 1), (1, 4), (4, 16),
                                   (93, 154), (256, 52), (52, 11), (11, 52), (17, 42), (256, 231), (128, 26)])
def test_resize_tables_match_oracle_bit_for_bit(sizes):
    from adunet_amd import resize_tables as rt
    from oracle import ops
    i, o = sizes
    s, w = rt.aa_spans(i, o)
    s2, w2 = ops.aa_triangle_spans(i, o)
    # the product drops the all-zero trailing tap column the nominal span carries at integer ratios
    assert np.array_equal(s, s2) and np.array_equal(w, w2[:, :w.shape[1]]) and not w2[:, w.shape[1]:].any()
    assert np.allclose(w.sum(1), 1.0, atol=1e-6)
    st, wt = rt.aa_spans_transposed(i, o)
    dense_t = np.zeros((i, o), np.float32)
    for j in range(i):
        for k in range(wt.shape[1]):
            if st[j] + k < o:
                dense_t[j, st[j] + k] += wt[j, k]
    assert np.array_equal(dense_t, ops.aa_matrix(i, o, np.float32).T)


def test_bucket_plan_covers_buffer_from_the_end():
    from adunet_amd.model import build_super_resolution_unet
    from adunet_amd.parallel import plan_buckets
    model, _ = build_super_resolution_unet(0.5, depth_override=3)
    total = model.count_params()
    buckets = plan_buckets(model.index, total, (8 << 20) // 4)
    assert buckets[0][1] == total and buckets[-1][0] == 0
    for (lo, hi), (lo2, hi2) in zip(buckets, buckets[1:]):
        assert hi2 == lo and lo2 < hi2
    offsets = {off for off, _ in model.index.values()}
    assert all(lo in offsets for lo, _ in buckets)
    assert sum(hi - lo for lo, hi in buckets) == total


def test_losses_contract():
    from adunet_amd.model import build_losses_and_metrics
    loss, metrics = build_losses_and_metrics("Charbonnier")
    assert loss.__name__ == "charbonnier_loss" and [m.__name__ for m in metrics] == ["psnr"]
    assert build_losses_and_metrics("l1")[0].__name__ == "l1_loss"
    with pytest.raises(ValueError, match="Unknown loss"):
        build_losses_and_metrics("mse")


def test_bench_starts_its_own_ranks_for_n_gpus(monkeypatch):
    """`python bench.py --gpus N` (no torchrun) must start N ranks itself: one child `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>` and its return code becomes ours;
    nothing in the parent touches the GPU before that."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        calls.append((cmd, env))
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    for var in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(var, raising=False)
    import torch

    def boom(*a, **k):
        raise AssertionError("the launching parent must not touch the GPU")

    monkeypatch.setattr(torch.cuda, "set_device", boom)
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 7
    (cmd, env), = calls
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"]
    assert cmd[-7].endswith("bench.py") and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_experiment_2_sweep_table_matches_the_reference_launcher():
    """Super_resolution/sbatch_scripts/run_experiment_adaptive_depth.sh:36-65: scales, depth per scale, run names, and the
    2080 Ti batch sizes (behind --reference_batch_sizes)."""
    from adunet_amd import run_experiment_adaptive_depth as R
    table = R.plan(reference_batch_sizes=True)
    assert [t["scale"] for t in table] == ["0.20", "0.30", "0.40", "0.50", "0.60", "0.70", "0.80"]
    assert [t["depth"] for t in table] == [1, 2, 3, 3, 4, 5, 5]
    assert [t["batch_size"] for t in table] == [8, 8, 6, 4, 3, 2, 1]
    assert table[3]["run_name"] == "exp2_adaptive_depth_scale0.50"
    assert all(t["batch_size"] >= r["batch_size"] for t, r in zip(R.plan(), table))
    args = R.parse_args(["--high_res_dir", "/x", "--scales", "0.30", "0.50"])
    assert args.scales == ["0.30", "0.50"] and args.patch_size == 256 and args.epochs == 100


def test_oracle_softmax_head_of_build_unet():
    """oracle restatement of build_unet(num_classes > 1) (unet_vinillia.py:89-90): class probabilities per pixel."""
    from oracle.seg_unet import SegUNetOracle
    o = SegUNetOracle(16, 8, 2, "ln", "convT", num_classes=4)
    assert o.param_shapes["mask_logits/kernel"] == (1, 1, 8, 4) and o.param_shapes["mask_logits/bias"] == (4,)
    rng = np.random.default_rng(0)
    params, state = o.init_params(rng)
    params["mask_logits/kernel"] = rng.standard_normal((1, 1, 8, 4))
    prob = o.forward(params, state, rng.random((2, 16, 16, 3)))
    assert prob.shape == (2, 16, 16, 4) and np.allclose(prob.sum(-1), 1.0) and (prob > 0).all()
    # one class: unchanged sigmoid head
    o1 = SegUNetOracle(16, 8, 2, "ln", "convT")
    assert o1.param_shapes["mask_logits/kernel"] == (1, 1, 8, 1)


def test_multitask_routing_follows_the_experiment_table():
    """Config 5's per-sample depth (adunet_amd.multitask): the Experiment-2 table of
    run_experiment_adaptive_depth.sh:47-55 where it has a row, the reference's heuristic elsewhere, clamped to 2..6."""
    from adunet_amd import multitask as M
    from adunet_amd.run_experiment_adaptive_depth import DEPTH_FOR_SCALE
    assert {f"{s:.2f}": d for s, d in M.EXPERIMENT2_DEPTH.items()} == DEPTH_FOR_SCALE
    assert [M.route_depth(s) for s in (0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8)] == [2, 2, 3, 3, 4, 5, 5]       # 0.2: table says 1, clamped
    assert M.route_depth(0.45) == 4 and M.route_depth(0.25) == 2 and M.route_depth(0.9) == 6               # heuristic, clamped
    with pytest.raises(ValueError):
        M.route_depth(1.5)
    rng = np.random.default_rng(0)
    samples = [(s, rng.random((8, 8, 3)), rng.random((8, 8, 3))) for s in (0.5, 0.3, 0.5, 0.6, 0.5)]
    got = [(k, a.shape[0]) for k, a, _ in M.bucket_by_depth(samples, batch_size=2)]
    assert got == [((0.5, 3), 2), ((0.5, 3), 1), ((0.3, 2), 1), ((0.6, 4), 1)]


def test_bank_gemm_tile_order_visits_every_tile_once_and_balances_the_xcds():
    """The XCD-aware tile order of the LDS-tiled bank GEMM (upconv.hip PlOrder; host twin ad_pw_gemm_tile_order): every tile of
    every (tiles_m, tiles_n, grid) exactly once, and the slowest workgroup within two rounds of the mean (before the leftover row
    groups were dealt block by block, the 145 x 18 tiles of Experiment 2's 34-wide level took 18 rounds on two XCDs, 9 on six)."""
    from adunet_amd import _lib
    lib = _lib.load()
    shapes = [(tm, tn, g) for tm in (1, 2, 3, 7, 8, 9, 31, 64, 125, 145, 146, 290, 1024) for tn in (1, 2, 3, 4, 6, 8, 9, 12, 18, 24, 36)
              for g in (256, 192, 64, 8, 5)]
    worst = 0.0
    for tm, tn, grid in shapes:
        g = min(grid, tm * tn)
        cap = 4 * (tm * tn + g - 1) // g + 64
        buf = np.full((g, cap), -2, np.int32)
        rounds = lib.ad_pw_gemm_tile_order(tm, tn, g, buf.ctypes.data, cap)
        assert rounds > 0, (tm, tn, g)
        seen = buf[buf >= 0]
        assert seen.size == tm * tn and np.array_equal(np.sort(seen), np.arange(tm * tn)), (tm, tn, g)
        if g % 8 == 0 and tm * tn >= 4 * g:
            worst = max(worst, rounds / (tm * tn / g))
            assert rounds <= np.ceil(tm * tn / g) * 1.5 + 2, (tm, tn, g, rounds)
    # the shape that showed the imbalance: 145 x 18 tiles on 256 workgroups = 10.2 rounds of work
    buf = np.empty((256, 64), np.int32)
    assert lib.ad_pw_gemm_tile_order(145, 18, 256, buf.ctypes.data, 64) <= 13
    # tile width: the 256-channel tile where it divides the width and does not cost rounds, else 192 / 128; 0 off the LDS path
    assert lib.ad_pw_gemm_tile_channels(8 * 63 * 63, 4608, 1024, _lib.AD_BF16) == 256
    assert lib.ad_pw_gemm_tile_channels(64 * 64 * 64, 128, 576, _lib.AD_BF16) == 192
    assert lib.ad_pw_gemm_tile_channels(64 * 64 * 64, 576, 128, _lib.AD_BF16) == 128
    assert lib.ad_pw_gemm_tile_channels(64 * 64 * 64, 128, 576, _lib.AD_F32) == 0
    lib.ad_set_option(b"no_pw_wide", 1)          # the A/B switch: no 256-channel tiles, whole row groups per XCD
    try:
        assert lib.ad_pw_gemm_tile_channels(8 * 63 * 63, 4608, 1024, _lib.AD_BF16) == 128
    finally:
        lib.ad_set_option(b"no_pw_wide", 0)
