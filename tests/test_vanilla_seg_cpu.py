"""CPU tests of the vanilla segmentation baseline's entry point (adunet_amd/seg_unet_vinillia.py, mirror of
Segmenation/code/unet_vinillia.py:101-297): pair discovery, the two tf.image.resize rules on closed forms, the dataset
contract, ReduceLROnPlateau's schedule and the epoch reduction of Keras' stateful metrics."""
import math
from pathlib import Path

import numpy as np
import pytest

from adunet_amd import seg_unet_vinillia as V
from adunet_amd.callbacks import ReduceLROnPlateau


def _folder(tmp_path, n=6, size=(30, 44), seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    img_dir, mask_dir = tmp_path / "img" / "sub", tmp_path / "mask"
    img_dir.mkdir(parents=True)
    mask_dir.mkdir()
    for i in (10, 2, 1, 33, 4, 5)[:n]:
        Image.fromarray(rng.integers(0, 256, size + (3,), dtype=np.uint8)).save(img_dir / f"ISIC_{i}.jpg")
        Image.fromarray((rng.random(size) > 0.6).astype(np.uint8) * 255).save(mask_dir / f"ISIC_{i}_segmentation.png")
    return tmp_path / "img", mask_dir


def test_discover_pairs_natural_order_keys_and_errors(tmp_path):
    img_dir, mask_dir = _folder(tmp_path)
    pairs = V._discover_pairs(img_dir, mask_dir, ".jpg", "_segmentation.png", None)
    assert [Path(p).stem for p, _ in pairs] == ["ISIC_1", "ISIC_2", "ISIC_4", "ISIC_5", "ISIC_10", "ISIC_33"]       # natural sort, recursive
    assert all(Path(m).stem == Path(p).stem + "_segmentation" for p, m in pairs)
    assert len(V._discover_pairs(img_dir, mask_dir, ".jpg", "_segmentation.png", 2)) == 2
    assert V._canonical_key(Path("Aachen_000001_gtFine_labelIds.png")) == "aachen_000001"
    assert V._canonical_key(Path("ISIC_0000123_Segmentation.PNG")) == "isic_0000123"
    with pytest.raises(ValueError, match="No images found"):
        V._discover_pairs(img_dir, mask_dir, ".bmp", "_segmentation.png", None)
    with pytest.raises(ValueError, match="No masks found"):
        V._discover_pairs(img_dir, mask_dir, ".jpg", "_nothing.png", None)
    (mask_dir / "ISIC_4_segmentation.png").unlink()
    with pytest.raises(ValueError, match="Missing mask for image ISIC_4.jpg"):
        V._discover_pairs(img_dir, mask_dir, ".jpg", "_segmentation.png", None)


def test_resize_rules_on_closed_forms():
    # bilinear, half-pixel centres, no antialias: identity at equal size; exact on a linear ramp away from the clamped border;
    # x2 enlargement of [0, 1] gives the 0.25 / 0.75 blend
    rng = np.random.default_rng(1)
    sq = rng.random((8, 8, 3)).astype(np.float32)
    assert np.allclose(V.resize_bilinear(sq, 8), sq, atol=1e-7)
    ramp = np.tile(np.arange(16, dtype=np.float32)[None, :, None], (16, 1, 1))
    out = V.resize_bilinear(ramp, 8)                       # centre of output i sits at input 2 i + 0.5
    assert np.allclose(out[0, :, 0], 2 * np.arange(8) + 0.5, atol=1e-6)
    two = np.array([[0.0, 1.0], [0.0, 1.0]], np.float32)[..., None]
    assert np.allclose(V.resize_bilinear(two, 4)[0, :, 0], [0.0, 0.25, 0.75, 1.0], atol=1e-7)
    # nearest, half-pixel centres: shrinking 6 -> 3 picks inputs 1, 3, 5; enlarging 2 -> 4 repeats each
    m = np.arange(36, dtype=np.float32).reshape(6, 6, 1)
    assert np.array_equal(V.resize_nearest(m, 3)[:, :, 0], m[1::2, 1::2, 0])
    assert np.array_equal(V.resize_nearest(two, 4)[0, :, 0], [0.0, 0.0, 1.0, 1.0])


def test_dataset_contract(tmp_path):
    img_dir, mask_dir = _folder(tmp_path)
    pairs = V._discover_pairs(img_dir, mask_dir, ".jpg", "_segmentation.png", None)
    ds = V.build_dataset(pairs, 16, 4, shuffle=True, augment=True, seed=13)
    b1 = list(ds)
    assert len(ds) == 2 and [b[0].shape for b in b1] == [(4, 16, 16, 3), (2, 16, 16, 3)] and b1[0][1].shape == (4, 16, 16, 1)
    assert all(b[0].dtype == np.float32 and 0 <= b[0].min() and b[0].max() <= 1 and set(np.unique(b[1])) <= {0.0, 1.0} for b in b1)
    b2 = list(ds)                                           # reshuffled every pass, reproducible from the seed
    assert not all(np.array_equal(x[0], y[0]) for x, y in zip(b1, b2))
    again = list(V.build_dataset(pairs, 16, 4, shuffle=True, augment=True, seed=13))
    assert all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(b1, again))
    plain = list(V.build_dataset(pairs, 16, 4, shuffle=False, augment=False, seed=13))
    img0, msk0 = V._parse_example(*pairs[0], 16)
    assert np.array_equal(plain[0][0][0], img0) and np.array_equal(plain[0][1][0], msk0)
    rng = np.random.default_rng(0)
    flips = {tuple(np.argwhere(V._augment(img0, msk0, rng)[0] == img0[0, 0, 0])[0][:2]) for _ in range(40)}
    assert {(0, 0), (0, 15), (15, 0), (15, 15)} == flips          # the four flip combinations all occur
    assert V.dice_coefficient(msk0, msk0) == pytest.approx(1.0) and V.dice_coefficient(msk0, 1 - msk0) < 1e-6


def test_reduce_lr_on_plateau_schedule():
    class Opt:
        learning_rate = 1e-4

    class M:
        optimizer = Opt()

    cb = ReduceLROnPlateau(monitor="val_loss", factor=0.5, patience=2, min_lr=3e-5)
    cb.set_model(M())
    cb.on_train_begin({})
    lrs = []
    for v in (1.0, 0.9, 0.95, 0.91, 0.899, 0.9, 0.9, 0.9, 0.9, 0.9, 0.9):     # improvements must exceed min_delta = 1e-4
        logs = {"val_loss": v}
        cb.on_epoch_end(len(lrs), logs)
        lrs.append(M.optimizer.learning_rate)
        assert "learning_rate" in logs
    # best 0.9 after epoch 2; epochs 3, 4 do not improve -> halve; 0.899 improves; then two more stalls -> halve; the floor
    assert lrs == pytest.approx([1e-4, 1e-4, 1e-4, 5e-5, 5e-5, 5e-5, 3e-5, 3e-5, 3e-5, 3e-5, 3e-5])
    with pytest.raises(ValueError):
        ReduceLROnPlateau(factor=1.0)
    M.optimizer.learning_rate = lambda step: 1e-3
    with pytest.raises(TypeError):
        cb.on_epoch_end(0, {"val_loss": 1.0})


def test_epoch_reduction_of_the_stateful_metrics():
    """Keras accumulates Precision / Recall / BinaryAccuracy as running sums over the epoch and means the loss and function
    metrics over the batches; SegModel._reduce_logs gets the summed per-batch tuples of _baseline_metrics_from."""
    from adunet_amd.seg_model import BASELINE_METRICS, SegModel
    m = SegModel.__new__(SegModel)
    m._baseline_metrics = True
    m.metrics_names = ["loss"] + list(BASELINE_METRICS)
    # two batches: (loss, acc, prec, rec, dice | correct, elems, tp, pp, pos)
    b1 = (0.5, 0.9, 1.0, 0.5, 0.6, 90.0, 100.0, 10.0, 10.0, 20.0)
    b2 = (0.3, 0.5, 0.25, 1.0, 0.2, 150.0, 300.0, 5.0, 20.0, 5.0)
    tot = [a + b for a, b in zip(b1, b2)]
    out = m._reduce_logs(m.metrics_names, tot, 2)
    assert out["loss"] == pytest.approx(0.4) and out["dice_coefficient"] == pytest.approx(0.4)
    assert out["accuracy"] == pytest.approx(240 / 400) and out["precision"] == pytest.approx(15 / 30) and out["recall"] == pytest.approx(15 / 25)
    assert list(out) == m.metrics_names
    zero = m._reduce_logs(m.metrics_names, [0.0] * 5 + [0.0, 10.0, 0.0, 0.0, 0.0], 1)
    assert zero["precision"] == 0.0 and zero["recall"] == 0.0                   # divide_no_nan
