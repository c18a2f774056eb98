#!/usr/bin/env python3
"""Offline evaluation entry point: mirror of /root/reference/Super_resolution/code/evaluate_model.py.

Same flags, same metric definitions (Y-channel PSNR / SSIM / MS-SSIM / MSE per patch, shave 2*round(1/scale)),
same report files (config.json, metrics.json, per_image_metrics.csv with the `<file>#patchNNNN` labels).
Checkpoints are the flat `.safetensors` / `.npz` files written by `Model.save_weights`.
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
from dataclasses import asdict, dataclass
from datetime import datetime
from pathlib import Path
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import metrics
from .metrics import infer_eval_shave  # noqa: F401  (re-exported, evaluate_model.py:49)
from .model import build_super_resolution_unet
from .pipeline import make_eval_patch_dataset, sorted_alphanumeric


@dataclass
class EvalResults:
    mse_mean: float
    mse_std: float
    psnr_mean: float
    psnr_std: float
    ssim_mean: float
    ssim_std: float
    msssim_mean: float
    msssim_std: float
    samples: int


def checkpoint_compute_dtype(model_path: Path):
    """The compute dtype a checkpoint was trained under (safetensors metadata written by Model.save_weights), or None."""
    import torch
    if str(model_path).endswith(".safetensors") and Path(model_path).exists():
        from safetensors import safe_open
        with safe_open(str(model_path), framework="numpy") as fh:
            name = (fh.metadata() or {}).get("compute_dtype")
        return {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}.get(name)
    return None


def load_checkpoint_model(model_path: Path, scale: float, patch_size: int, depth_override: int | None, dtype=None,
                          **build_kw):
    """evaluate_model.py:57-91 -- rebuild the architecture and load the weights.  Keras reloads a model under the
    policy it was saved with (float32 unless --mixed_precision): here the policy travels in the checkpoint metadata;
    `dtype` overrides it, and a checkpoint without metadata is evaluated in float32."""
    import torch
    if dtype is None:
        dtype = checkpoint_compute_dtype(model_path) or torch.float32
    model, _ = build_super_resolution_unet(scale=scale, input_size=patch_size, depth_override=depth_override, dtype=dtype,
                                           **build_kw)
    try:
        model.load_weights(str(model_path))
    except FileNotFoundError:
        raise
    except Exception as exc:
        raise RuntimeError(f"Failed to load weights from {model_path}: {exc}") from exc
    return model


def evaluate(model, dataset, eval_shave: int, with_ssim: bool = True) -> Tuple[EvalResults, List[Dict[str, float]]]:
    """evaluate_model.py:94-163.  The batch stays in HBM: forward, clip + BT.601 luma, shave (a strided window, no copy),
    per-image MSE / SSIM / MS-SSIM kernels (csrc/metrics.hip); only the per-image scalars come back to the host."""
    import torch
    model._require_device()
    dm = metrics.DeviceMetrics(model.device)
    vals = {"psnr": [], "ssim": [], "msssim": [], "mse": []}
    per_image: List[Dict[str, float]] = []
    offset = 0
    for lr_batch, hr_batch in dataset:
        pred = model(model._to_dev(lr_batch), training=False)                    # [B,P,P,3] fp32 on the device
        pred_y, hr_y = dm.luma(pred), dm.luma(model._to_dev(hr_batch))           # luma clips its input to [0, 1] first
        side = min(pred_y.shape[1:3]) - 2 * eval_shave
        mse, ssim, _ = dm.mse_ssim(hr_y, pred_y, shave=eval_shave, with_ssim=with_ssim)
        b_mse = mse.cpu().numpy()
        b_psnr = metrics.psnr_from_mse(b_mse)                                    # tf.image.psnr(max_val=1): inf at MSE 0
        b_ssim = ssim.cpu().numpy() if ssim is not None else np.full_like(b_psnr, np.nan)
        # MS-SSIM needs 5 halvings of an 11-pixel window
        b_ms = dm.msssim(hr_y, pred_y, shave=eval_shave) if with_ssim and side >= 11 * 16 else np.full_like(b_psnr, np.nan)
        for k, v in (("psnr", b_psnr), ("ssim", b_ssim), ("msssim", b_ms), ("mse", b_mse)):
            vals[k].append(v)
        for i in range(len(b_psnr)):
            per_image.append({"index": offset + i, "psnr_y": float(b_psnr[i]), "ssim_y": float(b_ssim[i]),
                              "msssim_y": float(b_ms[i]), "mse_y": float(b_mse[i])})
        offset += len(b_psnr)
    if not per_image:
        raise RuntimeError("Evaluation dataset yielded no samples.")

    return summarise({k: np.concatenate(v, axis=0) for k, v in vals.items()}), per_image


def summarise(columns: Dict[str, np.ndarray]) -> EvalResults:
    """evaluate_model.py:141-163: float64 mean / population std of the per-patch float32 columns `mse`, `psnr`, `ssim`,
    `msssim` (reproduces every metrics.json of the reference from its per_image_metrics.csv:
    tests/test_reference_metric_reports.py)."""
    (mse_m, mse_s), (p_m, p_s), (s_m, s_s), (ms_m, ms_s) = (metrics.aggregate(columns[k]) for k in ("mse", "psnr", "ssim", "msssim"))
    return EvalResults(mse_m, mse_s, p_m, p_s, s_m, s_s, ms_m, ms_s, int(len(columns["mse"])))


def attach_filenames(per_image: List[Dict[str, float]], filenames: Sequence[str]) -> None:
    if len(per_image) != len(filenames):
        raise ValueError("Per-image metric count does not match filename list.")
    for item, name in zip(per_image, filenames):
        item["filename"] = name


def write_outputs(run_dir: Path, summary: EvalResults, per_image, config: Dict[str, object], write_per_image: bool) -> None:
    run_dir.mkdir(parents=True, exist_ok=True)
    (run_dir / "config.json").write_text(json.dumps(config, indent=2))
    (run_dir / "metrics.json").write_text(json.dumps(asdict(summary), indent=2))
    if write_per_image:
        with (run_dir / "per_image_metrics.csv").open("w", newline="") as handle:
            writer = csv.DictWriter(handle, fieldnames=["index", "filename", "psnr_y", "ssim_y", "msssim_y", "mse_y"])
            writer.writeheader()
            for row in per_image:
                writer.writerow(row)


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="Evaluate a trained adaptive-depth U-Net checkpoint.")
    p.add_argument("--model-path", type=Path, required=True)
    p.add_argument("--scale", type=float, required=True)
    p.add_argument("--hr-dir", type=Path, required=True)
    p.add_argument("--patch-size", type=int, default=256)
    p.add_argument("--eval-stride", type=int, default=None)
    p.add_argument("--batch-size", type=int, default=8)
    p.add_argument("--limit", type=int, default=None)
    p.add_argument("--eval-shave", type=int, default=None)
    p.add_argument("--depth-override", type=int, default=None)
    p.add_argument("--output-dir", type=Path, default=Path("evaluation"))
    p.add_argument("--run-name", type=str, default=None)
    p.add_argument("--skip-per-image", action="store_true")
    p.add_argument("--dtype", choices=["float32", "bfloat16", "float16"], default=None,
                   help="compute dtype (default: the one recorded in the checkpoint, else float32)")
    p.add_argument("--mixed-precision", action="store_true", help="shorthand for --dtype float16 (the reference's policy)")
    return p.parse_args(argv)


def main(argv=None) -> None:
    args = parse_args(argv)
    hr_dir = Path(args.hr_dir).expanduser()
    if not hr_dir.exists():
        raise FileNotFoundError(f"High-resolution directory not found: {hr_dir}")
    hr_files = sorted_alphanumeric(glob.glob(str(hr_dir / "*.png")))
    if args.limit is not None and args.limit > 0:
        hr_files = hr_files[:args.limit]
    if not hr_files:
        raise ValueError(f"No high-resolution PNG files found in {hr_dir}")
    eval_ds, total, labels = make_eval_patch_dataset(hr_files, patch_size=args.patch_size, scale=args.scale,
                                                     batch_size=args.batch_size, stride=args.eval_stride)
    import torch
    dtype = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16, None: None}[args.dtype]
    if dtype is None and args.mixed_precision:
        dtype = torch.float16
    model = load_checkpoint_model(args.model_path.expanduser(), args.scale, args.patch_size, args.depth_override, dtype=dtype)
    shave = infer_eval_shave(args.scale, args.eval_shave)
    summary, per_patch = evaluate(model, eval_ds, eval_shave=shave)
    attach_filenames(per_patch, labels)
    print(f"Evaluated {summary.samples} patches ({len(hr_files)} images).")
    print(f"  PSNR(Y):     {summary.psnr_mean:.4f} ± {summary.psnr_std:.4f} dB")
    print(f"  SSIM(Y):     {summary.ssim_mean:.4f} ± {summary.ssim_std:.4f}")
    print(f"  MS-SSIM(Y):  {summary.msssim_mean:.4f} ± {summary.msssim_std:.4f}")
    print(f"  MSE(Y):      {summary.mse_mean:.6f} ± {summary.mse_std:.6f}")
    timestamp = datetime.now().strftime("%Y%m%d-%H%M%S")
    run_dir = Path(args.output_dir).expanduser() / (args.run_name or f"scale{args.scale:.2f}_{timestamp}")
    config = {"model_path": str(args.model_path.expanduser()), "scale": args.scale, "hr_dir": str(hr_dir),
              "patch_size": args.patch_size, "eval_stride": args.eval_stride or args.patch_size,
              "batch_size": args.batch_size, "limit": args.limit, "eval_shave": shave,
              "depth_override": args.depth_override, "samples": summary.samples, "images": len(hr_files),
              "compute_dtype": str(model.dtype).replace("torch.", ""),
              "created_at": timestamp}
    write_outputs(run_dir, summary, per_patch, config, write_per_image=not args.skip_per_image)
    print(f"[done] Report written to {run_dir}")


if __name__ == "__main__":
    main()
