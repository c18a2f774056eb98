#!/usr/bin/env python3
"""Training entry point: mirror of /root/reference/Super_resolution/code/train_adaptive_unet.py (`train(args)`,
`parse_args()`; same flags, same validation errors, same run artefacts: config.json, model_summary.txt, best
checkpoint named unet_adaptive_scale_new_loss{scale:.2f}_depth{d}, final Y-channel evaluation on val/test grids).
"""
from __future__ import annotations

import argparse
import glob
import json
import math
from datetime import datetime
from pathlib import Path

import numpy as np

from . import metrics
from .callbacks import BackupAndRestore, CSVLogger, EarlyStopping, ModelCheckpoint
from .evaluate_model import evaluate
from .model import (DEFAULT_BASE_CHANNELS, DEFAULT_RESIDUAL_HEAD_CHANNELS, Adam, build_losses_and_metrics,
                    build_super_resolution_unet)
from .pipeline import make_eval_patch_dataset, make_training_patch_dataset, sorted_alphanumeric, split_indices

DEFAULT_HR_SIZE = 256
DEFAULT_IMAGE_SUFFIX = ".png"
DATA_LR_SHRINK = 0.5            # train_adaptive_unet.py:60 -- training always degrades by x2


def train(args: argparse.Namespace):
    patch_size = args.patch_size
    if patch_size <= 0:
        raise ValueError("patch_size must be a positive integer.")
    if args.patches_per_image <= 0:
        raise ValueError("patches_per_image must be positive.")
    if args.eval_stride is not None and args.eval_stride <= 0:
        raise ValueError("eval_stride must be positive when provided.")
    if args.shuffle_buffer < 0:
        raise ValueError("shuffle_buffer must be non-negative.")
    if args.max_depth < 1:
        raise ValueError("max_depth must be at least 1.")
    if args.initial_epoch < 0:
        raise ValueError("initial_epoch must be non-negative.")
    if args.initial_epoch >= args.epochs:
        raise ValueError("initial_epoch must be smaller than --epochs to resume training.")
    if not args.high_res_dir:
        raise FileNotFoundError("High-resolution directory not found: (none given; pass --high_res_dir)")
    high_res_dir = Path(args.high_res_dir).expanduser()
    if not high_res_dir.exists():
        raise FileNotFoundError(f"High-resolution directory not found: {high_res_dir}")
    hr_paths = sorted_alphanumeric(glob.glob(str(high_res_dir / f"*{DEFAULT_IMAGE_SUFFIX}")))
    if args.limit and args.limit > 0:
        hr_paths = hr_paths[:args.limit]
    if not hr_paths:
        raise ValueError("No high-resolution images found with the given suffix.")
    train_split = 1.0 - (args.val_split + args.test_split)
    if train_split <= 0:
        raise ValueError("Validation and test splits leave no room for training data.")
    tr_idx, va_idx, te_idx = split_indices(len(hr_paths), train_split, args.val_split, args.test_split, args.seed)
    tr, va, te = ([hr_paths[i] for i in idx] for idx in (tr_idx, va_idx, te_idx))

    train_ds, train_count = make_training_patch_dataset(tr, patch_size, args.patches_per_image, DATA_LR_SHRINK,
                                                        args.batch_size, args.seed, args.shuffle_buffer)
    if getattr(args, "fast_feed", False):
        # MI355X feed path (not a reference flag): decode-once uint8 cache, forked crop workers, LR synthesis in HBM.  Started
        # here, before the model initialises the GPU in this process (the workers are forked and never touch HIP).
        from .pipeline import FastFeedDataset
        train_ds = FastFeedDataset(tr, patch_size, args.batch_size, DATA_LR_SHRINK, args.patches_per_image, args.seed,
                                   workers=args.feed_workers)
    val_ds = val_count = None
    if va:
        val_ds, val_count, _ = make_eval_patch_dataset(va, patch_size, DATA_LR_SHRINK, args.batch_size, stride=args.eval_stride)
    steps_per_epoch = math.ceil(train_count / args.batch_size)
    if steps_per_epoch <= 0:
        raise ValueError("Training dataset produced zero patches. Check patches_per_image or dataset splits.")

    import torch
    # --mixed_precision = the reference's mixed_float16 policy (:471-477): fp16 storage, fp32 variables, dynamic loss
    # scaling (compile() wraps Adam in a LossScaleOptimizer).  --bf16 is this build's throughput policy (no scaling needed).
    dtype = torch.bfloat16 if args.bf16 else torch.float16 if args.mixed_precision else torch.float32
    model, info = build_super_resolution_unet(args.scale, DEFAULT_BASE_CHANNELS, DEFAULT_RESIDUAL_HEAD_CHANNELS,
                                              depth_override=args.depth_override, input_size=patch_size,
                                              max_depth=args.max_depth, dtype=dtype, seed=args.seed)
    loss_fn, metric_fns = build_losses_and_metrics(args.loss)
    model.compile(optimizer=Adam(learning_rate=args.learning_rate), loss=loss_fn, metrics=metric_fns, jit_compile=False)

    if args.resume_from:
        resume = Path(args.resume_from).expanduser()
        if resume.is_dir():
            # the reference looks for its `.keras` archives (:500); this build's own checkpoints are `.safetensors`: newest of either
            cands = sorted(list(resume.glob("*.safetensors")) + list(resume.glob("*.keras")), key=lambda p: p.stat().st_mtime)
            if not cands:
                raise FileNotFoundError(f"No checkpoints found in {resume}")
            resume = cands[-1]
        if not resume.exists():
            raise FileNotFoundError(f"Checkpoint not found: {resume}")
        try:      # weights only, as the reference's model.load_weights (:511): the optimizer restarts.  --resume_optimizer also
            # brings back the Adam moments / iteration count / loss scaler when the checkpoint carries them
            model.load_weights(resume, restore_optimizer=bool(getattr(args, "resume_optimizer", False)))
        except Exception as exc:
            raise RuntimeError(f"Failed to load weights from {resume}: {exc}") from exc

    timestamp = datetime.now().strftime("%Y%m%d-%H%M%S")
    run_name = args.run_name or f"scale{args.scale:.2f}_depth{info['depth']}_{timestamp}"
    run_dir = Path(args.log_dir).expanduser() / run_name
    model_dir = Path(args.model_dir).expanduser()
    run_dir.mkdir(parents=True, exist_ok=True)
    ckpt = model_dir / f"unet_adaptive_scale_new_loss{args.scale:.2f}_depth{info['depth']}.safetensors"
    lines = []
    model.summary(print_fn=lines.append)
    (run_dir / "model_summary.txt").write_text("\n".join(lines))
    config = {**{k: (str(v) if isinstance(v, Path) else v) for k, v in vars(args).items()}, **info,
              "model_name": model.name, "params": model.count_params(), "train_images": len(tr), "val_images": len(va),
              "test_images": len(te), "steps_per_epoch": steps_per_epoch, "checkpoint": str(ckpt),
              "data_lr_shrink": DATA_LR_SHRINK, "compute_dtype": str(dtype)}
    (run_dir / "config.json").write_text(json.dumps(config, indent=2, default=str))

    monitor = "val_loss" if va else "loss"
    cbs = [EarlyStopping(monitor=monitor, patience=args.patience, restore_best_weights=True),
           ModelCheckpoint(ckpt, monitor=monitor, save_best_only=True), BackupAndRestore(run_dir / "train_backup"),
           CSVLogger(run_dir / "epoch_metrics.csv")]
    history = model.fit(train_ds, epochs=args.epochs, initial_epoch=args.initial_epoch, steps_per_epoch=steps_per_epoch,
                        validation_data=val_ds, callbacks=cbs, verbose=2)

    shave = metrics.infer_eval_shave(args.scale, args.eval_shave)
    if shave * 2 >= patch_size and patch_size > 0:                      # cap exactly as at :665-671
        adjusted = max(0, patch_size // 2 - 1)
        print(f"[warn] eval_shave={shave} removes the full frame for hr_size={patch_size}; reducing to {adjusted} pixels.")
        shave = adjusted
    final = {}
    for name, files in (("val", va), ("test", te)):
        if files:
            ds, _, _ = make_eval_patch_dataset(files, patch_size, DATA_LR_SHRINK, args.batch_size, stride=args.eval_stride)
            summary, _ = evaluate(model, ds, eval_shave=shave)
            final[name] = summary
            print(f"[eval:{name}] PSNR(Y) {summary.psnr_mean:.4f} dB  SSIM(Y) {summary.ssim_mean:.4f}  "
                  f"MS-SSIM(Y) {summary.msssim_mean:.4f}  MSE(Y) {summary.mse_mean:.6f}")
    return history, final


def parse_args(argv=None) -> argparse.Namespace:
    """Same flags as train_adaptive_unet.py:725-804."""
    p = argparse.ArgumentParser(description="Train adaptive-depth U-Net for super-resolution.")
    p.add_argument("--scale", type=float, required=True)
    p.add_argument("--batch_size", type=int, default=4)
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--loss", type=str, default="charbonnier", choices=["charbonnier", "l1", "combined"])
    p.add_argument("--patience", type=int, default=10)
    p.add_argument("--val_split", type=float, default=0.1)
    p.add_argument("--test_split", type=float, default=0.1)
    p.add_argument("--limit", type=int, default=None)
    p.add_argument("--seed", type=int, default=1234)
    p.add_argument("--patch_size", type=int, default=DEFAULT_HR_SIZE)
    p.add_argument("--patches_per_image", type=int, default=4)
    p.add_argument("--eval_stride", type=int, default=None)
    p.add_argument("--shuffle_buffer", type=int, default=1024)
    p.add_argument("--preview_patches", type=int, default=3)
    p.add_argument("--eval_shave", type=int, default=None)
    p.add_argument("--depth_override", type=int, default=None)
    p.add_argument("--max_depth", type=int, default=7)
    p.add_argument("--mixed_precision", action="store_true", help="mixed_float16 policy with dynamic loss scaling, as the reference")
    p.add_argument("--bf16", action="store_true", help="bf16 activations (MI355X throughput policy; not a reference flag)")
    p.add_argument("--model_dir", type=str, default="models")
    p.add_argument("--log_dir", type=str, default="logs")
    p.add_argument("--run_name", type=str, default=None)
    p.add_argument("--high_res_dir", type=str, default=None)
    p.add_argument("--low_res_dir", type=str, default=None)
    p.add_argument("--resume_from", type=str, default=None)
    p.add_argument("--resume_optimizer", action="store_true",
                   help="with --resume_from: also restore the optimizer state saved in the checkpoint (not a reference flag; the "
                        "reference reloads weights only)")
    p.add_argument("--fast_feed", action="store_true",
                   help="train from the decode-once / forked-worker / device-degradation feed (not a reference flag)")
    p.add_argument("--feed_workers", type=int, default=4)
    p.add_argument("--initial_epoch", type=int, default=0)
    return p.parse_args(argv)


if __name__ == "__main__":
    train(parse_args())
