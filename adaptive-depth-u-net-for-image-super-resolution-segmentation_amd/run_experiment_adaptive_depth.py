#!/usr/bin/env python3
"""Local (non-Slurm) launcher of the reference's Experiment 2 sweep -- adaptive depth per scale:
/root/reference/Super_resolution/sbatch_scripts/run_experiment_adaptive_depth.sh:36-97.

Same design table (scale -> encoder depth), same run names (`exp2_adaptive_depth_scale{scale}`), the same per-run
metadata file (scale, batch_size, depth, run_name, log_dir, model_dir, submitted) and the same extra arguments
(`--depth_override d --max_depth d`).  Instead of `sbatch` each run calls `train_adaptive_unet.train()` in this process,
one after the other; batch sizes default to MI355X-sized ones (the reference's were chosen for an 11 GB 2080 Ti) and can
be set back with --reference_batch_sizes.  Afterwards every run's checkpoint is scored with `evaluate_model` and one
`summary.csv` (scale, depth, params, PSNR / SSIM / MS-SSIM / MSE on Y) is written next to the metadata.
"""
from __future__ import annotations

import argparse
import csv
import glob
from datetime import datetime
from pathlib import Path

from . import evaluate_model, train_adaptive_unet
from .pipeline import make_eval_patch_dataset, sorted_alphanumeric

SCALES = ("0.20", "0.30", "0.40", "0.50", "0.60", "0.70", "0.80")
DEPTH_FOR_SCALE = {"0.20": 1, "0.30": 2, "0.40": 3, "0.50": 3, "0.60": 4, "0.70": 5, "0.80": 5}          # :47-55
REFERENCE_BATCH_SIZE = {"0.20": 8, "0.30": 8, "0.40": 6, "0.50": 4, "0.60": 3, "0.70": 2, "0.80": 1}     # :57-65
MI355X_BATCH_SIZE = {"0.20": 64, "0.30": 64, "0.40": 64, "0.50": 64, "0.60": 32, "0.70": 8, "0.80": 8}


def plan(scales=SCALES, reference_batch_sizes: bool = False):
    """The sweep table: one dict per run, in submission order."""
    sizes = REFERENCE_BATCH_SIZE if reference_batch_sizes else MI355X_BATCH_SIZE
    return [{"scale": s, "depth": DEPTH_FOR_SCALE.get(s, 3), "batch_size": sizes.get(s, 2),
             "run_name": f"exp2_adaptive_depth_scale{s}"} for s in scales]


def run(args: argparse.Namespace):
    base = Path(args.output_root).expanduser()
    log_base, model_base, meta_base = base / "logs" / "experiment_2", base / "models" / "Experiment_2", base / "metadata"
    for folder in (log_base, model_base, meta_base):
        folder.mkdir(parents=True, exist_ok=True)
    rows = []
    print("Submitting Experiment 2 runs (adaptive depth per scale)")
    for item in plan(args.scales or SCALES, args.reference_batch_sizes):
        stamp = datetime.now().strftime("%Y%m%d-%H%M%S")
        suffix = f"{item['run_name']}_{stamp}"
        log_dir, model_dir = log_base / suffix, model_base / suffix
        (meta_base / f"{suffix}.txt").write_text(
            f"scale={item['scale']}\nbatch_size={item['batch_size']}\ndepth={item['depth']}\nrun_name={item['run_name']}\n"
            f"log_dir={log_dir}\nmodel_dir={model_dir}\nsubmitted={datetime.now().astimezone().isoformat(timespec='seconds')}\n")
        print(f"  -> scale={item['scale']}, depth={item['depth']}, batch_size={item['batch_size']}, run_name={item['run_name']}")
        argv = ["--scale", item["scale"], "--batch_size", str(item["batch_size"]), "--depth_override", str(item["depth"]),
                "--max_depth", str(item["depth"]), "--log_dir", str(log_dir), "--model_dir", str(model_dir),
                "--run_name", item["run_name"], "--high_res_dir", args.high_res_dir, "--epochs", str(args.epochs),
                "--patch_size", str(args.patch_size), "--patches_per_image", str(args.patches_per_image),
                "--learning_rate", str(args.learning_rate), "--seed", str(args.seed)]
        if args.limit:
            argv += ["--limit", str(args.limit)]
        if args.bf16:
            argv.append("--bf16")
        if args.mixed_precision:
            argv.append("--mixed_precision")
        train_args = train_adaptive_unet.parse_args(argv)
        history, final = train_adaptive_unet.train(train_args)
        ckpts = sorted(model_dir.glob("*.safetensors"))
        row = {"scale": item["scale"], "depth": item["depth"], "batch_size": item["batch_size"], "run_name": item["run_name"],
               "epochs_ran": len(history.epoch), "checkpoint": str(ckpts[-1]) if ckpts else ""}
        if ckpts:                       # offline evaluation of the best checkpoint on the full image set (evaluate_model.py)
            files = sorted_alphanumeric(glob.glob(str(Path(args.high_res_dir).expanduser() / "*.png")))
            if args.limit:
                files = files[:args.limit]
            ds, _, _ = make_eval_patch_dataset(files, patch_size=args.patch_size, scale=float(item["scale"]),
                                               batch_size=item["batch_size"])
            model = evaluate_model.load_checkpoint_model(ckpts[-1], float(item["scale"]), args.patch_size, item["depth"])
            summary, _ = evaluate_model.evaluate(model, ds, eval_shave=evaluate_model.infer_eval_shave(float(item["scale"]), None))
            row.update(params=model.count_params(), psnr_y=summary.psnr_mean, ssim_y=summary.ssim_mean,
                       msssim_y=summary.msssim_mean, mse_y=summary.mse_mean, samples=summary.samples)
        rows.append(row)
    keys = ["scale", "depth", "batch_size", "run_name", "epochs_ran", "params", "psnr_y", "ssim_y", "msssim_y", "mse_y", "samples",
            "checkpoint"]
    with (meta_base / "summary.csv").open("w", newline="") as fh:
        writer = csv.DictWriter(fh, fieldnames=keys)
        writer.writeheader()
        for row in rows:
            writer.writerow({k: row.get(k, "") for k in keys})
    print(f"All Experiment 2 runs finished. Summary: {meta_base / 'summary.csv'}")
    return rows


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="Run the adaptive-depth-per-scale sweep (Experiment 2) locally.")
    p.add_argument("--high_res_dir", type=str, required=True)
    p.add_argument("--output_root", type=str, default="experiments/experiment_2_adaptive_depth")
    p.add_argument("--scales", type=str, nargs="*", default=None, help="subset of the design table, e.g. 0.30 0.50")
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("--patch_size", type=int, default=256)
    p.add_argument("--patches_per_image", type=int, default=4)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--seed", type=int, default=1234)
    p.add_argument("--limit", type=int, default=None)
    p.add_argument("--reference_batch_sizes", action="store_true", help="the 2080 Ti batch sizes of the reference's table")
    p.add_argument("--bf16", action="store_true")
    p.add_argument("--mixed_precision", action="store_true")
    return p.parse_args(argv)


if __name__ == "__main__":
    run(parse_args())
