"""CPU tests of the host data path and eval metrics (adunet_amd.pipeline / metrics): deterministic contracts of
/root/reference/shared/pipeline.py (sort order, grid patches, labels, split rules, RNG consumption) and metric
definitions.  cv2 pixel values for degrade_image are parity unpinned (OpenCV is not installed): only its
shape / dtype / range / resampling-identity properties are asserted."""
from pathlib import Path

import numpy as np
import pytest

from adunet_amd import metrics, pipeline
from oracle import metrics as ref_metrics
from oracle import ops as ref


def test_sorted_alphanumeric():
    names = ["img10.png", "img2.png", "IMG1.png", "a_12_b3.png", "a_2_b30.png"]
    assert pipeline.sorted_alphanumeric(names) == ["a_2_b30.png", "a_12_b3.png", "IMG1.png", "img2.png", "img10.png"]


def test_grid_patches_and_errors():
    img = np.arange(10 * 12 * 3, dtype=np.float32).reshape(10, 12, 3)
    p = pipeline.grid_patches(img, 4)
    assert p.shape == (6, 4, 4, 3) and np.array_equal(p[1], img[0:4, 4:8])
    assert pipeline.grid_patches(img, 4, stride=3).shape[0] == 3 * 3
    assert pipeline.grid_patches(img, 10, stride=50).shape[0] == 1
    with pytest.raises(ValueError):
        pipeline.grid_patches(img, 16)
    with pytest.raises(ValueError):
        pipeline.random_patch(img, 0)
    rng1, rng2 = np.random.default_rng(3), np.random.default_rng(3)
    a = pipeline.random_patches(img, 4, 5, rng=rng1)
    tops = [(int(rng2.integers(0, 7)), int(rng2.integers(0, 9))) for _ in range(5)]     # y first, then x
    assert all(np.array_equal(a[i], img[t:t + 4, l:l + 4]) for i, (t, l) in enumerate(tops))


def test_split_indices_rules():
    tr, va, te = pipeline.split_indices(100, 0.8, 0.1, 0.1, seed=1234)
    assert (len(tr), len(va), len(te)) == (80, 10, 10)
    assert sorted(np.concatenate([tr, va, te]).tolist()) == list(range(100))
    exp = np.arange(100)
    np.random.default_rng(1234).shuffle(exp)
    assert np.array_equal(tr, exp[:80])
    tr, va, te = pipeline.split_indices(3, 0.8, 0.1, 0.1, seed=0)
    assert len(tr) == 1 and len(va) + len(te) == 2
    with pytest.raises(ValueError):
        pipeline.split_indices(10, 1.0, 0.0, 0.0, seed=0)


def test_degrade_image_contract():
    rng = np.random.default_rng(0)
    hr = rng.random((32, 32, 3)).astype(np.float32)
    lr = pipeline.degrade_image(hr, 0.5, 32)
    assert lr.shape == (32, 32, 3) and lr.dtype == np.float32
    assert abs(float(lr.mean()) - float(hr.mean())) < 0.02          # area + cubic preserve the mean
    flat = np.full((16, 16, 3), 0.25, np.float32)
    assert np.allclose(pipeline.degrade_image(flat, 0.5, 16), 0.25, atol=1e-6)
    with pytest.raises(ValueError):
        pipeline.degrade_image(hr, 1.5, 32)


def test_eval_dataset_labels_and_batches(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    files = []
    for name, (h, w) in (("img2.png", (40, 72)), ("img10.png", (32, 32))):
        Image.fromarray((rng.random((h, w, 3)) * 255).astype(np.uint8)).save(tmp_path / name)
        files.append(str(tmp_path / name))
    files = pipeline.sorted_alphanumeric(files)
    ds, total, labels = pipeline.make_eval_patch_dataset(files, 32, 0.5, batch_size=2)
    assert total == 3 and labels == ["img2.png#patch0000", "img2.png#patch0001", "img10.png#patch0000"]
    batches = list(ds)
    assert [b[0].shape[0] for b in batches] == [2, 1] and batches[0][0].dtype == np.float32
    assert batches[0][1].min() >= 0 and batches[0][1].max() <= 1
    tds, n = pipeline.make_training_patch_dataset(files, 16, 3, 0.5, batch_size=4, seed=7, shuffle_buffer=4)
    it = iter(tds)
    lr, hr = next(it)
    assert n == 6 and lr.shape == (4, 16, 16, 3) and hr.shape == (4, 16, 16, 3)


def test_metrics_definitions():
    rng = np.random.default_rng(2)
    a = rng.random((2, 48, 48, 3)).astype(np.float32)
    assert np.allclose(metrics.rgb_to_luma_bt601(a), ref.rgb_to_luma_bt601(a.astype(np.float64)), atol=1e-6)
    assert [metrics.infer_eval_shave(s) for s in (0.5, 0.25, 0.6, 0.2, 0.8)] == [4, 8, 4, 10, 2]
    assert metrics.infer_eval_shave(0.5, 3) == 3 and ref.infer_eval_shave(0.3) == metrics.infer_eval_shave(0.3)
    y, b = a[..., :1], np.clip(a[..., :1] + 0.05 * rng.standard_normal((2, 48, 48, 1)).astype(np.float32), 0, 1)
    assert np.array_equal(metrics.rgb_to_luma_bt601(a), ref_metrics.rgb_to_luma_bt601(a))
    assert np.allclose(ref_metrics.psnr_per_image(y, b), ref.psnr_per_image(y.astype(np.float64), b.astype(np.float64)), atol=1e-4)
    assert np.isinf(ref_metrics.psnr_per_image(y, y)).all()
    mse = ref_metrics.mse_per_image(y, b)
    assert np.array_equal(metrics.psnr_from_mse(mse), ref_metrics.psnr_from_mse(mse))        # product's host form == oracle's
    s = ref_metrics.ssim_per_image(y, b)
    assert np.allclose(ref_metrics.ssim_per_image(y, y), 1.0) and (s < 1).all() and (s > 0.3).all()
    big = rng.random((1, 192, 192, 1)).astype(np.float32)
    assert np.allclose(ref_metrics.msssim_per_image(big, big), 1.0, atol=1e-6)
    noisy = np.clip(big + 0.1 * rng.standard_normal(big.shape).astype(np.float32), 0, 1)
    assert 0 < float(ref_metrics.msssim_per_image(big, noisy)[0]) < 1


def test_ssim_against_a_direct_2d_gaussian_filter():
    """tf.image.ssim restated twice: oracle/metrics.py filters separably; here the 11x11 sigma-1.5 window is applied as one 2-D
    VALID correlation (scipy.signal) and the SSIM map is formed from scratch."""
    from scipy.signal import correlate2d
    rng = np.random.default_rng(4)
    a = rng.random((2, 40, 37, 1))
    b = np.clip(a + 0.1 * rng.standard_normal(a.shape), 0, 1)
    x = np.arange(11) - 5.0
    g = np.exp(-x * x / (2 * 1.5 ** 2))
    g2 = np.outer(g, g)
    g2 /= g2.sum()
    want = []
    for i in range(2):
        p, q = a[i, :, :, 0], b[i, :, :, 0]
        f = lambda m: correlate2d(m, g2, mode="valid")
        mp, mq = f(p), f(q)
        vp, vq, cov = f(p * p) - mp * mp, f(q * q) - mq * mq, f(p * q) - mp * mq
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        want.append((((2 * mp * mq + c1) * (2 * cov + c2)) / ((mp * mp + mq * mq + c1) * (vp + vq + c2))).mean())
    assert np.allclose(ref_metrics.ssim_per_image(a, b), np.asarray(want), atol=1e-6)


# ----------------------------------------------------------------------------- ISIC data path of the segmentation trainer
def _isic_folder(tmp_path, n=5, size=40, with_mask=True, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    img_dir, mask_dir = tmp_path / "img", tmp_path / "mask"
    img_dir.mkdir(exist_ok=True)
    mask_dir.mkdir(exist_ok=True)
    for i in range(n):
        Image.fromarray((rng.random((size, size + 8, 3)) * 255).astype(np.uint8)).save(img_dir / f"ISIC_{i:07d}.jpg")
        if with_mask:
            m = np.zeros((size, size + 8), np.uint8)
            m[size // 4: size // 2 + i, size // 4: size // 2 + 2 * i] = 255
            Image.fromarray(m).save(mask_dir / f"ISIC_{i:07d}_segmentation.png")
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(img_dir / "ISIC_0000000_superpixels.png")   # must be ignored
    return img_dir, mask_dir


def test_collect_isic_pairs_contract(tmp_path):
    """Segmenation/code/train_adaptive_unet.py:70-135: pairing by ISIC id, superpixel files skipped, loud errors."""
    from adunet_amd import seg_train_adaptive_unet as T
    img_dir, mask_dir = _isic_folder(tmp_path)
    pairs = T.collect_isic_pairs(img_dir, mask_dir)
    assert len(pairs) == 5 and pairs[0][0].endswith("ISIC_0000000.jpg") and pairs[0][1].endswith("ISIC_0000000_segmentation.png")
    assert T.normalise_isic_key(Path("ISIC_0000123_Segmentation.PNG")) == "isic_0000123"
    with pytest.raises(FileNotFoundError):
        T.collect_isic_pairs(tmp_path / "nope", mask_dir)
    (mask_dir / "ISIC_0000003_segmentation.png").unlink()
    with pytest.raises(ValueError, match="Missing 1 segmentation masks"):
        T.collect_isic_pairs(img_dir, mask_dir)


def test_isic_loading_and_augmentation(tmp_path):
    from adunet_amd import seg_train_adaptive_unet as T
    img_dir, mask_dir = _isic_folder(tmp_path)
    pairs = T.collect_isic_pairs(img_dir, mask_dir)
    img, msk = T.load_isic_image(pairs[2][0], 32), T.load_isic_mask(pairs[2][1], 32)
    assert img.shape == (32, 32, 3) and img.dtype == np.float32 and 0.0 <= img.min() and img.max() <= 1.0
    assert msk.shape == (32, 32, 1) and set(np.unique(msk)) <= {0.0, 1.0} and msk.sum() > 0
    assert list(T._nearest_indices(4, 8)) == [0, 0, 1, 1, 2, 2, 3, 3] and list(T._nearest_indices(8, 4)) == [1, 3, 5, 7]
    assert np.allclose(T._bilinear_matrix(4, 8).sum(axis=1), 1.0)
    rng = np.random.default_rng(3)
    a_img, a_msk = T.apply_isic_augmentations(img, msk, 32, rng)
    assert a_img.shape == (32, 32, 3) and a_msk.shape == (32, 32, 1) and set(np.unique(a_msk)) <= {0.0, 1.0}
    # the same geometric transform hits image and mask: a mask painted into the image stays aligned
    marked = np.concatenate([msk, msk, msk], axis=-1)
    b_img, b_msk = T.apply_isic_augmentations(marked, msk, 32, np.random.default_rng(5))
    assert np.abs((b_img[..., :1] > 0.5).astype(np.float32) - b_msk).mean() < 0.05
    ds, count = T.build_isic_dataset(img_dir, mask_dir, batch_size=2, image_size=32, augment=True, shuffle=True, seed=1)
    batches = list(ds)
    assert count == 5 and len(ds) == 3 and [b[0].shape[0] for b in batches] == [2, 2, 1]
    first_pass = np.concatenate([b[1].reshape(b[1].shape[0], -1).sum(axis=1) for b in batches])
    second_pass = np.concatenate([b[1].reshape(b[1].shape[0], -1).sum(axis=1) for b in ds])
    assert not np.array_equal(first_pass, second_pass)               # reshuffled / re-augmented every pass


# ----------------------------------------------------------------------------- MI355X feed path (host half)
def test_banded_tables_reproduce_the_dense_resampling_matrices():
    for dense in (pipeline._area_matrix(64, 32), pipeline._area_matrix(50, 13), pipeline._cubic_matrix(32, 64),
                  pipeline._cubic_matrix(13, 50)):
        starts, weights = pipeline.banded_tables(dense)
        back = np.zeros_like(dense)
        for o in range(dense.shape[0]):
            back[o, starts[o]:starts[o] + weights.shape[1]] = weights[o]
        assert np.allclose(back, dense, atol=1e-7) and (starts >= 0).all() and (starts + weights.shape[1] <= dense.shape[1]).all()


def test_prefetch_loader_streams_crops_from_the_decoded_cache(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    files = []
    for i in range(3):
        arr = np.full((40, 56, 3), 10 * (i + 1), np.uint8)           # image i is the constant 10 (i + 1)
        arr[0, 0] = 255
        Image.fromarray(arr).save(tmp_path / f"im{i}.png")
        files.append(str(tmp_path / f"im{i}.png"))
    loader = pipeline.PrefetchPatchLoader(files, patch_size=16, batch_size=5, seed=3, workers=2, slots=4)
    try:
        seen = set()
        for _ in range(12):
            batch = next(loader)
            assert batch.shape == (5, 16, 16, 3) and batch.dtype == np.uint8
            seen |= {int(v) for v in np.unique(batch[:, 8, 8, 0])}
        assert seen <= {10, 20, 30} and len(seen) == 3              # crops of every image, never torn between images
    finally:
        loader.close()
    with pytest.raises(ValueError):
        pipeline.PrefetchPatchLoader(files, patch_size=64, batch_size=2)      # patch larger than the images
    # already decoded arrays (the synthetic in-memory set of `bench.py --feed loader`) are taken as they are
    arrays = [np.full((24, 24, 3), 7 * (i + 1), np.uint8) for i in range(2)]
    loader = pipeline.PrefetchPatchLoader(arrays, patch_size=16, batch_size=3, seed=1, workers=1, slots=2)
    try:
        assert {int(v) for _ in range(4) for v in np.unique(next(loader))} == {7, 14}
    finally:
        loader.close()
    with pytest.raises(ValueError):
        pipeline.PrefetchPatchLoader([np.zeros((24, 24), np.uint8)], patch_size=16, batch_size=2)


def test_croppers_follow_the_reference_contract():
    """grid_patches = row-major crops at multiples of the stride; random_patch consults the generator once per axis with
    room, rows first (shared/pipeline.py:97-174): a seeded stream must place the same crops as the reference's."""
    from adunet_amd import pipeline as P
    rng = np.random.default_rng(0)
    for h, w, p, st in [(17, 23, 5, None), (16, 16, 8, 3), (9, 30, 9, 7), (12, 12, 12, 5)]:
        img = rng.random((h, w, 3)).astype(np.float32)
        got = P.grid_patches(img, p, stride=st)
        step = st or p
        want = [img[t:t + p, l:l + p] for t in range(0, h - p + 1, step) for l in range(0, w - p + 1, step)]
        assert got.shape == (len(want), p, p, 3) and all(np.array_equal(a, b) for a, b in zip(got, want))
    img = rng.random((20, 9, 3)).astype(np.float32)
    a, b = np.random.default_rng(7), np.random.default_rng(7)
    for _ in range(5):
        crop = P.random_patch(img, 9, rng=a)                  # no room along the columns: ONE draw
        top = int(b.integers(0, 20 - 9 + 1))
        assert np.array_equal(crop, img[top:top + 9, :9])
    assert a.integers(0, 1 << 30) == b.integers(0, 1 << 30)   # the two generators are in the same state
    tr, va, te = P.split_indices(10, 0.7, 0.2, 0.1, seed=3)
    order = np.arange(10)
    np.random.default_rng(3).shuffle(order)
    assert np.array_equal(np.concatenate([tr, va, te]), order) and (len(tr), len(va), len(te)) == (7, 2, 1)
    with pytest.raises(ValueError, match="patch_size exceeds image dimensions"):
        P.grid_patches(img, 10)
    with pytest.raises(ValueError, match="stride must be positive"):
        P.grid_patches(img, 4, stride=-1)


def test_isic_dataset_cache_and_prefetch_give_the_same_stream(tmp_path):
    """Decode-once cache + background prefetch (the tf.data map(AUTOTUNE) + prefetch of Segmenation/code/
    train_adaptive_unet.py:214-225) must not change what the stream yields: same batches as the uncached, unthreaded
    iteration, pass after pass (pass k is seeded seed + k), enlargement by AREA like every other size."""
    from PIL import Image
    from adunet_amd import seg_train_adaptive_unet as S
    rng = np.random.default_rng(0)
    (tmp_path / "img").mkdir()
    (tmp_path / "msk").mkdir()
    for i, (h, w) in enumerate([(40, 52), (33, 33), (20, 64), (70, 18), (48, 48)]):     # two of them smaller than the target
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(tmp_path / "img" / f"ISIC_{i:07d}.png")
        Image.fromarray((rng.random((h, w)) > 0.5).astype(np.uint8) * 255).save(tmp_path / "msk" / f"ISIC_{i:07d}_segmentation.png")
    pairs = S.collect_isic_pairs(tmp_path / "img", tmp_path / "msk")
    fast = S.IsicDataset(pairs, 2, 32, augment=True, shuffle=True, seed=5)
    slow = S.IsicDataset(pairs, 2, 32, augment=True, shuffle=True, seed=5, cache=False, prefetch=0)
    for _ in range(3):                                   # three passes: the second and third come from the cache
        a, b = list(fast), list(slow)
        assert len(a) == len(b) == 3 and a[-1][0].shape[0] == 1
        for (ia, ma), (ib, mb) in zip(a, b):
            assert np.array_equal(ia, ib) and np.array_equal(ma, mb) and ia.dtype == np.float32 and set(np.unique(ma)) <= {0.0, 1.0}
    img = S.load_isic_image(pairs[3][0], 32)             # 70 x 18 -> 32 x 32: shrink one axis, enlarge the other, AREA both
    assert img.shape == (32, 32, 3) and 0.0 <= img.min() and img.max() <= 1.0
    # a consumer that abandons a pass (evaluate(steps=...) breaks out, an exception in the step) must not leave
    # the producer thread blocked in put() for ever; the next pass still yields the full, correct stream
    import threading
    before = threading.active_count()
    for _ in range(4):
        it = iter(S.IsicDataset(pairs, 1, 32, augment=False, shuffle=False, seed=5, prefetch=1))
        next(it)
        it.close()                                       # what a `break` out of a for loop does to the generator
    assert threading.active_count() <= before
    again = S.IsicDataset(pairs, 2, 32, augment=True, shuffle=True, seed=5)
    first = next(iter(again))                            # abandoned pass 0 ...
    full = list(again)                                   # ... pass 1 is complete and equals the unthreaded stream's pass 1
    ref_ds = S.IsicDataset(pairs, 2, 32, augment=True, shuffle=True, seed=5, cache=False, prefetch=0)
    list(ref_ds)
    want = list(ref_ds)
    assert len(full) == 3 and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(full, want))
    # a first iterator that is still REFERENCED (neither closed nor collected) while a second pass starts must not
    # dead-lock the second pass (its producer used to hold the cache lock for the whole pass while blocked on its full queue)
    held = S.IsicDataset(pairs, 1, 32, augment=False, shuffle=False, seed=5, prefetch=1)
    it1 = iter(held)
    first_item = next(it1)                               # producer 1 is now blocked in put(): queue of 1 is full
    done = []
    t = threading.Thread(target=lambda: done.append(list(held)), daemon=True)
    t.start()
    t.join(timeout=20.0)
    assert not t.is_alive() and len(done) == 1 and len(done[0]) == 5        # the second pass ran to completion
    with pytest.raises((RuntimeError, StopIteration)):   # the superseded iterator ends instead of hanging
        for _ in range(10):
            next(it1)
    assert first_item[0].shape == (1, 32, 32, 3)
