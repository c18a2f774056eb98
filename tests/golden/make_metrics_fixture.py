"""Condense the reference's committed evaluation reports into one NPZ + JSON fixture.

Run in the build container (reads /root/reference as *data*; nothing is imported or executed):
    python tests/golden/make_metrics_fixture.py
Inputs:  Super_resolution/experiments/*/evaluation/*/{per_image_metrics.csv,metrics.json,config.json}
         (written by the reference's evaluate_model.py:173-190 on the author's GPU).
Outputs: tests/golden/eval_reports.npz   -- per run one float32 array [3598, 4] = (psnr_y, ssim_y, msssim_y, mse_y);
                                            the CSV cells are float(float32) reprs, so float32 holds them exactly
                                            (checked below), and the patch labels of the first run
         tests/golden/eval_reports.json  -- per run: metrics.json verbatim (inf / nan kept as strings), the scale,
                                            eval_shave, depth_override and sample / image counts of config.json,
                                            and whether every run lists the same labels in the same order.
These are the only floating-point outputs of the reference in its tree: they pin tf.image.psnr's float32 arithmetic
(psnr_y from mse_y), the behaviour of all four metrics on a degenerate patch (the `inf,1.0,1.0,0.0` row) and the
float64 mean / population-std aggregation of evaluate_model.py:141-163.
"""
import csv
import glob
import json
import math
import os

import numpy as np

ROOT = "/root/reference/Super_resolution/experiments"
HERE = os.path.dirname(os.path.abspath(__file__))
COLS = ("psnr_y", "ssim_y", "msssim_y", "mse_y")


def jsonable(v):
    if isinstance(v, float) and (math.isinf(v) or math.isnan(v)):
        return repr(v)            # 'inf' / 'nan' : strict JSON has no spelling for them
    return v


def main():
    arrays, meta, labels0, same_labels = {}, {}, None, True
    for d in sorted(glob.glob(os.path.join(ROOT, "*", "evaluation", "*"))):
        key = os.path.relpath(d, ROOT).replace(os.sep, "|")
        with open(os.path.join(d, "per_image_metrics.csv"), newline="") as fh:
            rows = list(csv.DictReader(fh))
        vals64 = np.array([[float(r[c]) for c in COLS] for r in rows], dtype=np.float64)
        vals32 = vals64.astype(np.float32)
        # lossless: every cell is the repr of a float32
        assert np.array_equal(vals32.astype(np.float64), vals64, equal_nan=True), key
        assert [int(r["index"]) for r in rows] == list(range(len(rows))), key
        labels = [r["filename"] for r in rows]
        if labels0 is None:
            labels0 = labels
        same_labels = same_labels and labels == labels0
        arrays[key] = vals32
        metrics = json.load(open(os.path.join(d, "metrics.json")))        # Python's json reads Infinity / NaN
        cfg = json.load(open(os.path.join(d, "config.json")))
        meta[key] = {"source": os.path.relpath(d, "/root/reference"),
                     "metrics": {k: jsonable(v) for k, v in metrics.items()},
                     "config": {k: cfg.get(k) for k in ("scale", "eval_shave", "depth_override", "samples", "images",
                                                        "patch_size", "eval_stride", "batch_size")}}
    np.savez_compressed(os.path.join(HERE, "eval_reports.npz"), labels=np.array(labels0), **arrays)
    with open(os.path.join(HERE, "eval_reports.json"), "w") as fh:
        json.dump({"columns": list(COLS), "same_labels_in_every_run": same_labels, "runs": meta}, fh, indent=1, sort_keys=True)
    print(f"{len(arrays)} runs, {sum(a.shape[0] for a in arrays.values())} rows, same labels: {same_labels}")


if __name__ == "__main__":
    main()
