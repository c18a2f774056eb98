"""The reference's own numeric outputs as pins (CPU).

`Super_resolution/experiments/*/evaluation/*/{per_image_metrics.csv,metrics.json}` are the only floating-point results of
the reference in its tree: 15 evaluation runs x 3 598 DIV2K-valid patches, written by evaluate_model.py:173-190.  They are
condensed to tests/golden/eval_reports.{npz,json} by tests/golden/make_metrics_fixture.py (data, read as text).  They pin:
  (i)   tf.image.psnr's float32 arithmetic: `psnr_y` from `mse_y` (evaluate_model.py:118,121), incl. `inf` at MSE 0;
  (ii)  the aggregation of evaluate_model.py:141-163 (float64 mean, population std): every field of every metrics.json;
  (iii) the report schema: 3 598 rows, labels `<file>#patchNNNN` in natural order (shared/pipeline.py:285), shave rule.
The oracle (oracle/metrics.py) and the product's host code (adunet_amd/metrics.py, evaluate_model.summarise) are both held
to them.  The `-m gpu` counterpart of the degenerate row lives in tests/test_metrics_gpu.py.
"""
import json
import math
import os
import re

import numpy as np
import pytest

from adunet_amd import evaluate_model, metrics, pipeline
from oracle import metrics as ref_metrics

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def reports():
    z = np.load(os.path.join(GOLDEN, "eval_reports.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "eval_reports.json")))
    assert meta["columns"] == ["psnr_y", "ssim_y", "msssim_y", "mse_y"]
    runs = {k: z[k] for k in z.files if k != "labels"}
    assert len(runs) == 15 and set(runs) == set(meta["runs"])
    return runs, meta, [str(s) for s in z["labels"]]


def ulp_distance(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


@pytest.mark.parametrize("impl", [ref_metrics.psnr_from_mse, metrics.psnr_from_mse], ids=["oracle", "product"])
def test_psnr_column_follows_from_the_mse_column_in_float32(reports, impl):
    runs, _, _ = reports
    rows = exact = 0
    for key, a in runs.items():
        psnr, mse = a[:, 0], a[:, 3]
        got = impl(mse)
        assert got.dtype == np.float32
        zero = mse == 0
        assert np.array_equal(np.isinf(psnr), zero) and np.isinf(got[zero]).all() and (got[zero] > 0).all(), key
        d = ulp_distance(got[~zero], psnr[~zero])
        # float32 rounding: TensorFlow's own log kernel and its second evaluation of the mean leave <= 2 units in the last place
        assert d.max() <= 2, (key, int(d.max()))
        rows += d.size
        exact += int((d == 0).sum())
    assert rows == 15 * 3598 - 2                # two runs (scale 0.20) hold the all-black patch
    assert exact / rows > 0.985, exact / rows
    # the form this replaced, -10 * log10(mse), is NOT what the reference ran: most cells differ
    a = runs[sorted(runs)[0]]
    fin = a[:, 3] > 0
    assert (ulp_distance((-10.0 * np.log10(a[fin, 3])).astype(np.float32), a[fin, 0]) > 0).mean() > 0.5


def test_degenerate_row_of_the_reference(reports):
    """exp1_depth3_scale0.20_eval/per_image_metrics.csv:1388 = `1386,0839.png#patch0000,inf,1.0,1.0,0.0`."""
    runs, meta, labels = reports
    hits = [k for k in runs if np.isinf(runs[k][:, 0]).any()]
    assert sorted(meta["runs"][k]["config"]["scale"] for k in hits) == [0.2, 0.2]
    for k in hits:
        i = int(np.flatnonzero(np.isinf(runs[k][:, 0]))[0])
        assert i == 1386 and labels[i] == "0839.png#patch0000"
        assert runs[k][i].tolist() == [float("inf"), 1.0, 1.0, 0.0]
    # the oracle on identical planes (any constant, and a textured one): exactly that row
    rng = np.random.default_rng(0)
    for plane in (np.full((1, 236, 236, 1), 16 / 255, np.float32), rng.random((1, 236, 236, 1)).astype(np.float32)):
        row = [float(ref_metrics.psnr_per_image(plane, plane)[0]), float(ref_metrics.ssim_per_image(plane, plane)[0]),
               float(ref_metrics.msssim_per_image(plane, plane)[0]), float(ref_metrics.mse_per_image(plane, plane)[0])]
        assert row == [float("inf"), 1.0, 1.0, 0.0], row


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_every_metrics_json_follows_from_its_csv(reports, impl):
    runs, meta, _ = reports
    for key, a in runs.items():
        want = {k: float(v) if isinstance(v, str) else v for k, v in meta["runs"][key]["metrics"].items()}
        cols = {"psnr": a[:, 0], "ssim": a[:, 1], "msssim": a[:, 2], "mse": a[:, 3]}
        if impl == "oracle":
            got = {"samples": len(a)}
            for name, col in cols.items():
                got[f"{name}_mean"], got[f"{name}_std"] = ref_metrics.aggregate(col)
        else:
            from dataclasses import asdict
            got = asdict(evaluate_model.summarise(cols))
        assert set(got) == set(want), key
        for field, w in want.items():
            g = got[field]
            if isinstance(w, float) and math.isnan(w):
                assert math.isnan(g), (key, field)
            elif isinstance(w, float) and math.isinf(w):
                assert g == w, (key, field)
            else:
                assert g == pytest.approx(w, rel=1e-12, abs=0), (key, field, g, w)


def test_report_schema_labels_and_shave(reports):
    runs, meta, labels = reports
    assert meta["same_labels_in_every_run"]
    assert len(labels) == 3598 and all(a.shape == (3598, 4) for a in runs.values())
    assert all(re.fullmatch(r"\d{4}\.png#patch\d{4}", s) for s in labels)
    files = []
    for s in labels:
        f, idx = s.split("#patch")
        if not files or files[-1][0] != f:
            files.append([f, 0])
        assert int(idx) == files[-1][1]              # patch indices count up from 0 inside each file
        files[-1][1] += 1
    names = [f for f, _ in files]
    assert len(names) == 100 and names == pipeline.sorted_alphanumeric(names)       # natural sort, 0801 ... 0900
    for key, m in meta["runs"].items():
        cfg = m["config"]
        assert cfg["samples"] == 3598 and cfg["images"] == 100 and cfg["patch_size"] == 256
        # evaluate_model.py:49-54 -- shave = 2 * round(1 / scale), for the product's host code and the oracle's
        assert metrics.infer_eval_shave(cfg["scale"]) == cfg["eval_shave"] == ref_metrics.infer_eval_shave(cfg["scale"]), key


def test_patch_counts_per_file_are_grid_counts_of_2k_images():
    """shared/pipeline.py:139-174 -- a stride grid WITHOUT an edge-flush tile: a 2 040-pixel side gives
    (2040 - 256) // 256 + 1 = 7 crops.  The images' sizes are not in the reference's tree; what the labels do pin is that every
    per-file count is 7 x rows with 3 <= rows <= 7 (the other side between 768 and 2 040 pixels), which an edge-flush grid
    (8 per 2 040 pixels) could not produce, and the product's grid_patches yields exactly such counts."""
    z = np.load(os.path.join(GOLDEN, "eval_reports.npz"))
    counts = {}
    for s in z["labels"]:
        counts[str(s).split("#")[0]] = counts.get(str(s).split("#")[0], 0) + 1
    assert sum(counts.values()) == 3598
    for f, n in counts.items():
        assert n % 7 == 0 and 3 <= n // 7 <= 7, (f, n)
    most = max(set(counts.values()), key=list(counts.values()).count)
    assert most == 35                                             # 2 040 x 1 356 (3:2), the commonest DIV2K format
    img = np.zeros((1356, 2040, 3), np.float32)
    assert pipeline.grid_patches(img, 256).shape[0] == 35
    assert pipeline.grid_patches(np.zeros((2040, 2040, 3), np.float32), 256).shape[0] == 49
