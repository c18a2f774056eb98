"""GPU: the reference's entry points end to end on synthetic PNGs -- train(args) (+resume, early stopping,
checkpoint, run artefacts) then evaluate_model.main() on the checkpoint it wrote."""
import json

import numpy as np
import pytest
import torch

from conftest import free_port

pytestmark = pytest.mark.gpu


def _pngs(folder, n, size=48, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    for i in range(n):
        base = rng.random((size // 4, size // 4, 3))
        img = np.kron(base, np.ones((4, 4, 1)))                     # piecewise-constant: something to super-resolve
        Image.fromarray((img * 255).astype(np.uint8)).save(folder / f"img{i}.png")


def test_train_then_evaluate(device, tmp_path):
    from adunet_amd import evaluate_model, train_adaptive_unet as T
    hr = tmp_path / "hr"
    hr.mkdir()
    _pngs(hr, 10)
    argv = ["--scale", "0.5", "--high_res_dir", str(hr), "--patch_size", "32", "--depth_override", "1", "--batch_size", "4",
            "--epochs", "2", "--patches_per_image", "2", "--learning_rate", "1e-3", "--model_dir", str(tmp_path / "models"),
            "--log_dir", str(tmp_path / "logs"), "--run_name", "t", "--bf16", "--shuffle_buffer", "8"]
    history, final = T.train(T.parse_args(argv))
    assert history.epoch == [0, 1] and "val_loss" in history.history
    run = tmp_path / "logs" / "t"
    cfg = json.loads((run / "config.json").read_text())
    assert cfg["depth"] == 1 and cfg["params"] == 520003 and cfg["model_name"] == "U-Net_SR_scale0.50_depth1"
    assert (run / "model_summary.txt").exists() and (run / "epoch_metrics.csv").read_text().startswith("epoch,")
    ckpt = tmp_path / "models" / "unet_adaptive_scale_new_loss0.50_depth1.safetensors"
    assert ckpt.exists() and set(final) == {"val", "test"}
    evaluate_model.main(["--model-path", str(ckpt), "--scale", "0.5", "--hr-dir", str(hr), "--patch-size", "32",
                         "--depth-override", "1", "--output-dir", str(tmp_path / "eval"), "--run-name", "e", "--batch-size", "4"])
    m = json.loads((tmp_path / "eval" / "e" / "metrics.json").read_text())
    assert m["samples"] == 10 and m["psnr_mean"] > 10
    rows = (tmp_path / "eval" / "e" / "per_image_metrics.csv").read_text().splitlines()
    assert rows[0] == "index,filename,psnr_y,ssim_y,msssim_y,mse_y" and rows[1].split(",")[1] == "img0.png#patch0000"
    # the same weights as a Keras-3 `.keras` archive (what the reference's ModelCheckpoint writes and evaluate_model / --resume_from
    # read, :57-91 / train :496-522; written and read without h5py; store naming restated, container pinned): identical evaluation report, and a
    # directory holding only the archive resumes from it
    from adunet_amd.evaluate_model import load_checkpoint_model
    keras_dir = tmp_path / "keras_models"
    keras_dir.mkdir()
    load_checkpoint_model(ckpt, 0.5, 32, 1).save(keras_dir / "unet_adaptive_scale_new_loss0.50_depth1.keras")
    for name, path in (("e_st", ckpt), ("e_keras", keras_dir / "unet_adaptive_scale_new_loss0.50_depth1.keras")):
        evaluate_model.main(["--model-path", str(path), "--scale", "0.5", "--hr-dir", str(hr), "--patch-size", "32", "--dtype", "bfloat16",
                             "--depth-override", "1", "--output-dir", str(tmp_path / "eval"), "--run-name", name, "--batch-size", "4"])
    assert (tmp_path / "eval" / "e_st" / "metrics.json").read_text() == (tmp_path / "eval" / "e_keras" / "metrics.json").read_text()
    h3, _ = T.train(T.parse_args(argv + ["--resume_from", str(keras_dir), "--initial_epoch", "1"]))
    assert h3.epoch == [1]
    # resume path and argument validation
    argv2 = argv + ["--resume_from", str(tmp_path / "models"), "--initial_epoch", "1"]
    h2, _ = T.train(T.parse_args(argv2))
    assert h2.epoch == [1]
    with pytest.raises(ValueError):
        T.train(T.parse_args(argv + ["--initial_epoch", "5"]))
    with pytest.raises(FileNotFoundError):
        T.train(T.parse_args(["--scale", "0.5", "--high_res_dir", str(tmp_path / "nope")]))


def test_fp32_checkpoint_is_evaluated_in_fp32(device, tmp_path):
    """The offline evaluation reloads a model under the policy it was trained with (Keras behaviour,
    evaluate_model.py:57-91): an fp32-trained checkpoint must give the fp32 model's PSNR (1e-3 dB), not a bf16 re-run."""
    import torch
    from adunet_amd import evaluate_model, train_adaptive_unet as T
    from adunet_amd.pipeline import make_eval_patch_dataset, sorted_alphanumeric
    hr = tmp_path / "hr"
    hr.mkdir()
    _pngs(hr, 6)
    argv = ["--scale", "0.5", "--high_res_dir", str(hr), "--patch_size", "32", "--depth_override", "1", "--batch_size", "4",
            "--epochs", "1", "--patches_per_image", "2", "--learning_rate", "1e-3", "--model_dir", str(tmp_path / "models"),
            "--log_dir", str(tmp_path / "logs"), "--run_name", "t32", "--shuffle_buffer", "8"]
    T.train(T.parse_args(argv))
    ckpt = tmp_path / "models" / "unet_adaptive_scale_new_loss0.50_depth1.safetensors"
    assert evaluate_model.checkpoint_compute_dtype(ckpt) == torch.float32
    common = ["--model-path", str(ckpt), "--scale", "0.5", "--hr-dir", str(hr), "--patch-size", "32", "--depth-override", "1",
              "--output-dir", str(tmp_path / "eval"), "--batch-size", "4"]
    evaluate_model.main(common + ["--run-name", "auto"])
    cfg = json.loads((tmp_path / "eval" / "auto" / "config.json").read_text())
    assert cfg["compute_dtype"] == "float32"
    got = json.loads((tmp_path / "eval" / "auto" / "metrics.json").read_text())["psnr_mean"]
    files = sorted_alphanumeric([str(f) for f in hr.glob("*.png")])
    ds, _, _ = make_eval_patch_dataset(files, patch_size=32, scale=0.5, batch_size=4)
    model = evaluate_model.load_checkpoint_model(ckpt, 0.5, 32, 1, dtype=torch.float32)
    want, _ = evaluate_model.evaluate(model, ds, eval_shave=evaluate_model.infer_eval_shave(0.5, None))
    assert abs(got - want.psnr_mean) < 1e-3
    evaluate_model.main(common + ["--run-name", "mp", "--mixed-precision"])
    cfg = json.loads((tmp_path / "eval" / "mp" / "config.json").read_text())
    assert cfg["compute_dtype"] == "float16"
    evaluate_model.main(common + ["--run-name", "bf", "--dtype", "bfloat16"])
    assert json.loads((tmp_path / "eval" / "bf" / "config.json").read_text())["compute_dtype"] == "bfloat16"


def test_segmentation_train_entry_point(device, tmp_path):
    """Segmenation/code/train_adaptive_unet.py:463-575 end to end on a synthetic ISIC-shaped folder: protocol B, run
    artefacts with the reference's config keys, best-val_dice checkpoint, backup removed after a completed fit."""
    from PIL import Image
    from adunet_amd import seg_train_adaptive_unet as T
    rng = np.random.default_rng(0)
    folders = {}
    for split, n in (("train", 6), ("val", 3)):
        for kind in ("img", "mask"):
            folders[(split, kind)] = tmp_path / f"{split}_{kind}"
            folders[(split, kind)].mkdir()
        for i in range(n):
            m = np.zeros((48, 48), np.uint8)
            m[8 + i: 30, 10: 28 + i] = 255
            img = (rng.random((48, 48, 3)) * 80 + m[..., None] * 0.5).astype(np.uint8)      # the lesion is brighter
            Image.fromarray(img).save(folders[(split, "img")] / f"ISIC_{i:07d}.jpg")
            Image.fromarray(m).save(folders[(split, "mask")] / f"ISIC_{i:07d}_segmentation.png")
    argv = ["--protocol", "B", "--epochs", "2", "--batch_size", "3", "--depth", "2", "--image_size", "32", "--bf16",
            "--train_images", str(folders[("train", "img")]), "--train_masks", str(folders[("train", "mask")]),
            "--val_images", str(folders[("val", "img")]), "--val_masks", str(folders[("val", "mask")]),
            "--model_dir", str(tmp_path / "models"), "--log_dir", str(tmp_path / "logs"), "--run_name", "segrun"]
    history, metrics = T.train(T.parse_args(argv))
    assert history.epoch == [0, 1] and {"loss", "dice", "iou", "val_dice"} <= set(history.history)
    cfg = json.loads((tmp_path / "logs" / "segrun" / "config.json").read_text())
    assert cfg["protocol"] == "B" and cfg["batch_size"] == 3 and cfg["train_samples"] == 6 and cfg["val_samples"] == 3
    assert cfg["train_steps_per_epoch"] == 2 and cfg["initial_lr"] == 3e-4 and set(cfg["metrics"]) == {"loss", "dice", "iou"}
    assert cfg["model_name"] == "adaptive_unet_depth2_c64"
    assert (tmp_path / "models" / "segrun.safetensors").exists() and (tmp_path / "logs" / "segrun" / "model_summary.txt").exists()
    assert not (tmp_path / "logs" / "segrun" / "train_backup" / "backup.safetensors").exists()
    with pytest.raises(FileNotFoundError):
        T.train(T.parse_args(["--protocol", "A"]))


@pytest.mark.parametrize("policy", [[], ["--dtype", "bfloat16"], ["--mixed_precision"]], ids=["float32", "bfloat16", "mixed_float16"])
def test_vanilla_segmentation_baseline_entry_point(device, tmp_path, policy):
    """Segmenation/code/unet_vinillia.py:236-293 end to end on a synthetic folder: BinaryCrossentropy, the four Keras metrics
    (accuracy / precision / recall as running sums over the epoch, dice_coefficient as a batch mean), checkpoints on
    val_dice_coefficient, ReduceLROnPlateau's learning-rate log; and one batch's metric values against NumPy on the model's
    own probabilities."""
    from PIL import Image
    from adunet_amd import seg_unet_vinillia as V
    rng = np.random.default_rng(0)
    folders = {}
    for split, n in (("train", 6), ("val", 3)):
        for kind in ("img", "mask"):
            folders[(split, kind)] = tmp_path / f"{split}_{kind}"
            folders[(split, kind)].mkdir()
        for i in range(n):
            m = np.zeros((48, 48), np.uint8)
            m[8 + i: 30, 10: 28 + i] = 255
            img = (rng.random((48, 48, 3)) * 80 + m[..., None] * 0.5).astype(np.uint8)
            Image.fromarray(img).save(folders[(split, "img")] / f"ISIC_{i:07d}.jpg")
            Image.fromarray(m).save(folders[(split, "mask")] / f"ISIC_{i:07d}_segmentation.png")
    argv = ["--epochs", "2", "--batch_size", "4", "--depth", "2", "--image_size", "32", "--base_channels", "32", "--augment",
            "--train_image_dir", str(folders[("train", "img")]), "--train_mask_dir", str(folders[("train", "mask")]),
            "--val_image_dir", str(folders[("val", "img")]), "--val_mask_dir", str(folders[("val", "mask")]),
            "--model_dir", str(tmp_path / "models"), "--run_name", "vanilla", "--fit_verbose", "0"] + policy
    model, history = V.train(V.parse_args(argv))
    want_dtype = {"": torch.float32, "--dtype": torch.bfloat16, "--mixed_precision": torch.float16}[policy[0] if policy else ""]
    assert model.dtype == want_dtype
    if policy == ["--mixed_precision"]:          # Keras wraps the optimizer under mixed_float16; ReduceLROnPlateau reaches the inner rate
        assert type(model.optimizer).__name__ == "LossScaleOptimizer" and model.optimizer.inner_optimizer.learning_rate == 1e-4
    keys = {"loss", "accuracy", "precision", "recall", "dice_coefficient"}
    assert history.epoch == [0, 1] and keys | {"val_" + k for k in keys} | {"learning_rate"} <= set(history.history)
    assert history.history["learning_rate"] == [1e-4, 1e-4]
    assert all(0.0 <= history.history[k][-1] <= 1.0 for k in keys - {"loss"})
    assert (tmp_path / "models" / "vanilla_best.safetensors").exists() and (tmp_path / "models" / "vanilla_final.safetensors").exists()
    assert model.name == "unet_isic_baseline" and model.metrics_names == ["loss", "accuracy", "precision", "recall", "dice_coefficient"]
    # one batch against NumPy on the probabilities the model itself returns
    val = V.build_dataset(V._discover_pairs(folders[("val", "img")], folders[("val", "mask")], ".jpg", "_segmentation.png", None),
                          32, 3, shuffle=False, augment=False, seed=0)
    img, mask = next(iter(val))
    got = [float(v) for v in model.test_on_batch(img, mask)]
    p = model(img).astype(np.float64)
    pos, truth = p > 0.5, mask > 0.5
    pc = np.clip(p, 1e-7, 1 - 1e-7)
    want = [float(np.mean(-(mask * np.log(pc) + (1 - mask) * np.log(1 - pc)))), float(np.mean(pos == truth)),
            float((pos & truth).sum() / max(pos.sum(), 1)), float((pos & truth).sum() / max(truth.sum(), 1)),
            V.dice_coefficient(mask, p)]
    assert np.allclose(got[:5], want, rtol=2e-4, atol=2e-5), (got[:5], want)       # (on the model's OWN probabilities: any dtype)
    assert got[5:] == [float((pos == truth).sum()), float(mask.size), float((pos & truth).sum()), float(pos.sum()), float(truth.sum())]
    res = model.evaluate(val, return_dict=True)
    assert list(res) == model.metrics_names and res["accuracy"] == pytest.approx(want[1], abs=1e-6)
    with pytest.raises(FileNotFoundError):
        V.train(V.parse_args([]))
    with pytest.raises(ValueError, match="unknown metrics"):
        model.compile(optimizer=None, loss=V.binary_crossentropy(), metrics=["accuracy", "auc"])


def test_bench_plain_and_under_torchrun_agree(device):
    """The driver starts N = 1 as `python bench.py --gpus 1 ...` and N > 1 through torch.distributed.run: both start-up
    paths must run and report the same workload (throughput within 5 %; at world size 1 the second goes through RCCL,
    the bucketed all-reduce and the segmented graph replay)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = ["--gpus", "1", "--steps", "8", "--warmup", "3", "--no-cpu-baseline", "--no-micro"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    plain = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + flags, capture_output=True, text=True, timeout=600,
                           env=env, cwd=root)
    assert plain.returncode == 0, plain.stderr[-2000:]
    dist_run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                               "127.0.0.1", "--master-port", str(free_port()), os.path.join(root, "bench.py")] + flags,
                              capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert dist_run.returncode == 0, dist_run.stderr[-2000:]
    # stdout is the ONE JSON line and nothing else (RCCL's version banner, written to file descriptor 1 when the first
    # communicator is created, is sent to stderr by bench.py)
    assert len(plain.stdout.strip().splitlines()) == 1 and len(dist_run.stdout.strip().splitlines()) == 1, dist_run.stdout[:2000]
    a = json.loads(plain.stdout.strip().splitlines()[-1])
    b = json.loads(dist_run.stdout.strip().splitlines()[-1])
    assert a["metric"] == b["metric"] and a["config"]["workload"] == b["config"]["workload"] and a["n_gpus"] == b["n_gpus"] == 1
    assert b["rccl_ranks"] == 1 and b["exposed_comm_ms_per_step"] is not None and "rccl_ranks" not in a
    assert abs(a["value"] - b["value"]) < 0.05 * a["value"], (a["value"], b["value"])
    for line in (a, b):
        rf = line["roofline"]
        assert {"bound", "family", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_step", "families", "hbm_ops",
                "share_of_step", "hbm_gb_per_step", "hbm_frac", "executed_gflop_per_step"} <= set(rf)
        assert 0.2 < rf["frac"] < 1.0 and 0.1 < rf["frac_step"] < 1.0
        # the line names the conv family with the largest share of the step
        assert rf["family"] == max(rf["families"], key=lambda k: rf["families"][k]["ms_per_step"])
        assert rf["executed_gflop_per_step"] < rf["algorithmic_gflop_per_step"]      # the factored up-convs
        # both roofs per family: the LayerNorm-forward launches move their operand bytes at 3-6 TB/s (the HBM side of the ridge)
        assert all({"hbm_tb_per_s", "flop_per_byte", "share_of_step"} <= set(f) for f in rf["families"].values())
        assert 2.5 < rf["families"]["fused_ln_fwd"]["hbm_tb_per_s"] < 8.0
        # r04: traffic and operand bytes of the named family are quoted over the same launches; the executed-FLOP step fraction
        # sits below the reference-graph one (factored up-convs); the r03 rocm-smi power model is gone from the line
        assert rf["traffic_ops"] and "power_model" not in rf and rf["frac_step_executed"] < rf["frac_step"]
        # the roof that binds the named family by the roofline model itself (arithmetic intensity against the 312.5 FLOP/B ridge)
        br = rf["binding_roof"]
        assert br["bound"] == ("hbm" if br["flop_per_byte"] < br["ridge_flop_per_byte"] else "mfma") and 0.2 < br["frac_of_8_tb_per_s"] < 1.0
        # secondary figure: fractions against the peak at the in-kernel clock (profiles/r*_inkernel_clock.json), when committed
        if rf["inkernel_clock_source"]:
            for name in ("fused_ln_fwd", "wgrad", "dgrad_ln_bwd_fused"):
                f = rf["families"][name]
                assert 1.2 < f["inkernel_clock_ghz"] <= 2.45 and f["frac"] <= f["frac_at_clock"] < 1.0, (name, f)


def test_bench_starts_two_gloo_ranks_on_one_gpu(device):
    """`python bench.py --gpus 2 --backend gloo`: the N > 1 start-up path of the driver's scaling run (bench.py starts its own
    ranks as a child `torch.distributed.run` before anything in the parent touches the GPU) rehearsed on the one-GPU box:
    two ranks share the device, gradients travel over gloo.  Rank 0 prints exactly ONE JSON line, the child's return code
    becomes ours, both process groups are destroyed (the command returns)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3",
                          "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-micro"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = run.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), run.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks"] == 2 and rec["dist_backend"] == "gloo" and "rccl_ranks" not in rec
    assert rec["config"]["global_batch"] == 16 and rec["config"]["parallelism"] == "dp2"
    assert rec["exposed_comm_ms_per_step"] is not None and rec["exchange"] == "torch.distributed.all_reduce"
    assert rec["value"] > 0 and np.isfinite(rec["config"]["final_loss"])
    # a run that fails inside the ranks (zero timed steps: rank 0 cannot form a rate): the children's return code comes back
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "0",
                          "--warmup", "0", "--batch", "2", "--workload", "K1", "--no-cpu-baseline", "--no-micro"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]


def test_bench_config5_stream_alone_and_under_two_gloo_ranks(device):
    """`bench.py --workload K5` (BASELINE config 5, build-defined): one JSON line with the mixed stream's items; and the same under
    two ranks sharing the GPU over gloo -- every model of the bank with its own DataParallel (`AdaptiveDepthBank.data_parallel`)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra, ranks in ((["--batch", "8"], 1), (["--batch", "4", "--gpus", "2", "--backend", "gloo"], 2)):
        run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "K5", "--steps", "2", "--warmup", "1"] + extra,
                             capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert run.returncode == 0, run.stderr[-3000:]
        lines = run.stdout.strip().splitlines()
        assert len(lines) == 1, run.stdout[-2000:]
        rec = json.loads(lines[0])
        items = rec["config"]["items"]
        assert rec["dtype"] == "f16" and rec["n_gpus"] == ranks and "BUILD-DEFINED" in rec["config"]["workload"]
        assert [(i["task"], i["scale"], i["depth"]) for i in items] == [("sr", 0.3, 2), ("sr", 0.5, 3), ("sr", 0.6, 4), ("sr", 0.7, 5),
                                                                        ("seg", None, 4)]
        assert rec["value"] > 0 and all(np.isfinite(i["final_loss"]) and i["images_per_s"] > 0 for i in items)
        assert rec["config"]["global_batch"] == sum(i["batch"] for i in items) * ranks
        if ranks == 2:
            assert rec["ranks"] == 2 and rec["dist_backend"] == "gloo"


def test_bench_feed_loader_reports_the_fed_rate(device):
    """`bench.py --feed loader`: the resident-batch figure stays `value`; beside it the same graph-replayed
    step fed by the host data path (crop workers -> shared-memory ring -> uint8 over PCIe -> LR synthesis in HBM), with the H2D
    bytes per step (uint8 HR crops only: B x P x P x 3)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--feed", "loader", "--feed-workers", "2", "--steps", "6",
                          "--warmup", "2", "--batch", "8", "--no-cpu-baseline", "--no-micro"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert run.returncode == 0, run.stderr[-3000:]
    rec = json.loads(run.stdout.strip().splitlines()[-1])
    feed = rec["feed"]
    assert feed["h2d_bytes_per_step"] == 8 * 256 * 256 * 3 and feed["images_per_s"] > 0 and feed["loader_alone_images_per_s"] > 0
    assert 0.05 < feed["fraction_of_resident_rate"] < 1.2 and rec["value"] > 0 and np.isfinite(rec["config"]["final_loss"])


def test_experiment_sweep_launcher(device, tmp_path):
    """Two rows of the Experiment-2 table end to end: per-run metadata files with the reference's keys, checkpoints,
    and the summary table with the offline Y-channel metrics."""
    from adunet_amd import run_experiment_adaptive_depth as R
    hr = tmp_path / "hr"
    hr.mkdir()
    _pngs(hr, 8, size=64)
    rows = R.run(R.parse_args(["--high_res_dir", str(hr), "--output_root", str(tmp_path / "exp"), "--scales", "0.30", "0.50",
                               "--epochs", "1", "--patch_size", "32", "--patches_per_image", "2", "--learning_rate", "1e-3",
                               "--reference_batch_sizes", "--bf16"]))
    assert [r["depth"] for r in rows] == [2, 3] and all(r["psnr_y"] > 5 and r["samples"] > 0 for r in rows)
    metas = sorted((tmp_path / "exp" / "metadata").glob("exp2_adaptive_depth_scale*.txt"))
    assert len(metas) == 2 and set(l.split("=")[0] for l in metas[0].read_text().splitlines()) == {
        "scale", "batch_size", "depth", "run_name", "log_dir", "model_dir", "submitted"}
    table = (tmp_path / "exp" / "metadata" / "summary.csv").read_text().splitlines()
    assert table[0].startswith("scale,depth,batch_size,run_name") and len(table) == 3
