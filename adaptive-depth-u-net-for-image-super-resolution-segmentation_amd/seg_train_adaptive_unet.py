#!/usr/bin/env python3
"""Segmentation training entry point: mirror of /root/reference/Segmenation/code/train_adaptive_unet.py
(`train(args)` :463-575, `parse_args()` :583-610, the ISIC-2017 data path :70-256, `prepare_callbacks` :410-448).

Same flags, protocols (A / B), run artefacts (`config.json` with the reference's keys, `model_summary.txt`, best
checkpoint on `val_dice`, `train_backup/`), same pairing / validation errors.  The host data path is NumPy + Pillow
instead of tf.data + tf.image: area shrink of the image, nearest-neighbour shrink of the mask (half-pixel centres),
rot90 / flips / 1.00-1.15 zoom + random crop drawn from one `numpy.random.Generator(seed)`.  TensorFlow's random
streams cannot be reproduced, so the augmentation is distribution-equivalent, not draw-for-draw; tf.image pixel
values are parity-unpinned (TensorFlow is not installable here), the deterministic parts are tested on the CPU.
"""
from __future__ import annotations

import argparse
import json
import math
from datetime import datetime
from pathlib import Path
from typing import Iterator, List, Sequence, Tuple

import numpy as np

from .callbacks import BackupAndRestore, CSVLogger, EarlyStopping, ModelCheckpoint
from .pipeline import _area_matrix
from .seg_model import (DEFAULT_BASE_CHANNELS, DEFAULT_DEPTH, DEFAULT_IMAGE_SIZE, PROTOCOLS, build_adaptive_depth_unet,
                        build_optimizer)

DEFAULT_SEED = 42
DEFAULT_THRESHOLD = 0.5
IMAGE_SUFFIXES = {".jpg", ".jpeg", ".png"}
MASK_SUFFIXES = {".png", ".jpg"}


# ----------------------------------------------------------------------------- pairing (:70-135)
def normalise_isic_key(path: Path) -> str:
    """Lower-case stem without the `_segmentation` token: the key images and masks share."""
    return Path(path).stem.lower().replace("_segmentation", "")


def collect_isic_pairs(image_dir, mask_dir) -> List[Tuple[str, str]]:
    """(image, mask) path pairs sorted by ISIC id.  FileNotFoundError for a missing / empty directory, ValueError
    listing (up to five of) the images without a mask; `*superpixels*` files are not images."""
    image_dir, mask_dir = Path(image_dir), Path(mask_dir)
    for what, folder in (("Image", image_dir), ("Mask", mask_dir)):
        if not folder.exists():
            raise FileNotFoundError(f"{what} directory does not exist: {folder}")
    images = sorted((p for p in image_dir.iterdir()
                     if p.is_file() and p.suffix.lower() in IMAGE_SUFFIXES and "superpixels" not in p.stem.lower()),
                    key=lambda p: p.stem.lower())
    masks = {normalise_isic_key(p): p for p in sorted(mask_dir.iterdir(), key=normalise_isic_key)
             if p.is_file() and p.suffix.lower() in MASK_SUFFIXES and p.stem.lower().endswith("_segmentation")}
    if not images:
        raise FileNotFoundError(f"No image files found in {image_dir}")
    if not masks:
        raise FileNotFoundError(f"No mask files found in {mask_dir}")
    orphans = [p.name for p in images if normalise_isic_key(p) not in masks]
    if orphans:
        raise ValueError(f"Missing {len(orphans)} segmentation masks in {mask_dir}; examples: "
                         f"{', '.join(orphans[:5])}{'…' if len(orphans) > 5 else ''}")
    return [(str(p), str(masks[normalise_isic_key(p)])) for p in images]


# ----------------------------------------------------------------------------- decoding / resizing (:138-157)
def _nearest_indices(n_in: int, n_out: int) -> np.ndarray:
    """tf.image.resize(NEAREST_NEIGHBOR), half-pixel centres: source index floor((i + 0.5) * n_in / n_out)."""
    idx = np.floor((np.arange(n_out) + 0.5) * (n_in / n_out)).astype(np.int64)
    return np.minimum(idx, n_in - 1)


def _bilinear_matrix(n_in: int, n_out: int) -> np.ndarray:
    """tf.image.resize(BILINEAR, antialias=False): half-pixel centres, edge clamp."""
    m = np.zeros((n_out, n_in), np.float64)
    src = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
    lo = np.floor(src).astype(np.int64)
    frac = src - lo
    for o in range(n_out):
        m[o, min(max(lo[o], 0), n_in - 1)] += 1.0 - frac[o]
        m[o, min(max(lo[o] + 1, 0), n_in - 1)] += frac[o]
    return m


def _resize_separable(img: np.ndarray, my: np.ndarray, mx: np.ndarray) -> np.ndarray:
    return np.einsum("pw,owc->opc", mx, np.einsum("oh,hwc->owc", my, img.astype(np.float64)))


def load_isic_image(path, size: int) -> np.ndarray:
    """RGB float32 [size, size, 3] in [0, 1], area-resized (tf.image.ResizeMethod.AREA)."""
    from PIL import Image
    with Image.open(str(path)) as im:
        rgb = np.asarray(im.convert("RGB"), np.float32) / 255.0
    h, w = rgb.shape[:2]
    # tf.image.ResizeMethod.AREA in both directions, as the reference (:146): each output pixel averages the input area
    # it covers (for an enlargement that area lies inside one or two input pixels)
    return _resize_separable(rgb, _area_matrix(h, size), _area_matrix(w, size)).astype(np.float32)


def load_isic_mask(path, size: int) -> np.ndarray:
    """Binary float32 [size, size, 1]: nearest-neighbour resize, then > 0.5."""
    from PIL import Image
    with Image.open(str(path)) as im:
        g = np.asarray(im.convert("L"), np.float32) / 255.0
    g = g[_nearest_indices(g.shape[0], size)][:, _nearest_indices(g.shape[1], size)]
    return (g > 0.5).astype(np.float32)[..., None]


# ----------------------------------------------------------------------------- augmentation (:160-196)
def apply_isic_augmentations(image: np.ndarray, mask: np.ndarray, size: int, rng: np.random.Generator):
    """rot90 by a random quarter turn, independent left-right / up-down flips, zoom by U[1, 1.15) (bilinear for the
    image, nearest for the mask), random crop back to `size`; the mask is re-binarised."""
    flip_lr, flip_ud = rng.random() > 0.5, rng.random() > 0.5
    turns = int(rng.integers(0, 4))

    def geometric(t):
        t = np.rot90(t, turns, axes=(0, 1))
        t = t[:, ::-1] if flip_lr else t
        return t[::-1] if flip_ud else t

    image, mask = geometric(image), geometric(mask)
    zoomed = int(round(float(rng.uniform(1.0, 1.15)) * size))
    if zoomed != size:
        m = _bilinear_matrix(size, zoomed)
        image = _resize_separable(image, m, m).astype(np.float32)
        idx = _nearest_indices(size, zoomed)
        mask = mask[idx][:, idx]
    top = int(rng.integers(0, zoomed - size + 1))
    left = int(rng.integers(0, zoomed - size + 1))
    image = np.ascontiguousarray(image[top:top + size, left:left + size])
    mask = np.ascontiguousarray(mask[top:top + size, left:left + size])
    return image.astype(np.float32), (mask > 0.5).astype(np.float32)


class IsicDataset:
    """Re-iterable stream of (image [B,S,S,3], mask [B,S,S,1]) float32 batches (build_isic_dataset, :199-226):
    optional full reshuffle per pass, optional augmentation, last batch kept (drop_remainder=False).

    The reference maps decode + resize over tf.data with AUTOTUNE parallelism and prefetch (:214-225).  Here every file
    is decoded and area-resized ONCE (a multi-megapixel ISIC photograph costs ~10 GFLOP of float64 resize) into an
    in-memory cache at `image_size` -- float32 images as the reference's pipeline holds them, uint8 masks -- and a pass
    only augments cached arrays.  Batches are assembled by a background thread `prefetch` batches ahead of the consumer
    (NumPy releases the GIL in the resize / copy kernels), so the GPU step and the host augmentation overlap."""

    def __init__(self, pairs: Sequence[Tuple[str, str]], batch_size: int, image_size: int, augment: bool, shuffle: bool,
                 seed: int, cache: bool = True, prefetch: int = 4):
        self.pairs, self.batch_size, self.size = list(pairs), int(batch_size), int(image_size)
        self.augment, self.shuffle, self.seed = augment, shuffle, seed
        self._pass = 0
        self._cache = {} if cache else None
        self.prefetch = int(prefetch)
        import threading
        self._cache_lock = threading.Lock()

    def __len__(self) -> int:
        return math.ceil(len(self.pairs) / self.batch_size)

    def _item(self, i: int) -> Tuple[np.ndarray, np.ndarray]:
        # the lock covers ONE item's cache read / fill (never a whole pass: a producer that is blocked on a full queue
        # must not keep the next pass's producer out -- ADVICE r04)
        with self._cache_lock:
            if self._cache is not None and i in self._cache:
                img, msk8 = self._cache[i]
                return img, msk8.astype(np.float32)
        img, msk = load_isic_image(self.pairs[i][0], self.size), load_isic_mask(self.pairs[i][1], self.size)
        if self._cache is not None:
            with self._cache_lock:
                self._cache[i] = (img, msk.astype(np.uint8))
        return img, msk

    def _batches(self, rng: np.random.Generator, order: np.ndarray) -> Iterator[Tuple[np.ndarray, np.ndarray]]:
        imgs, masks = [], []
        for i in order:
            img, msk = self._item(int(i))
            if self.augment:
                img, msk = apply_isic_augmentations(img, msk, self.size, rng)
            imgs.append(img)
            masks.append(msk)
            if len(imgs) == self.batch_size:
                yield np.stack(imgs), np.stack(masks)
                imgs, masks = [], []
        if imgs:
            yield np.stack(imgs), np.stack(masks)

    def __iter__(self) -> Iterator[Tuple[np.ndarray, np.ndarray]]:
        rng = np.random.default_rng(self.seed + self._pass)          # reshuffle_each_iteration=True
        self._pass += 1
        order = rng.permutation(len(self.pairs)) if self.shuffle else np.arange(len(self.pairs))
        source = self._batches(rng, order)
        if self.prefetch <= 0:
            yield from source
            return
        import queue
        import threading
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        done = object()
        stop = threading.Event()       # set when the consumer leaves (break, exception, garbage-collected generator)
        # one active pass per dataset: a new iteration ends the previous one's producer even if its generator is still referenced
        prev = getattr(self, "_active_stop", None)
        if prev is not None:
            prev.set()
        self._active_stop = stop

        def put(item) -> bool:
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def produce():
            try:
                for item in source:
                    if not put(item):
                        return
                put(done)
            except BaseException as exc:        # surfaces in the consumer
                put(exc)

        worker = threading.Thread(target=produce, daemon=True)
        worker.start()
        try:
            while True:
                try:
                    item = q.get(timeout=0.5)
                except queue.Empty:
                    if stop.is_set() and not worker.is_alive():      # superseded by a newer pass over this dataset
                        raise RuntimeError("IsicDataset: this iteration was ended by a newer iteration over the same dataset")
                    continue
                if item is done:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:            # also runs on generator close(): the producer stops instead of blocking in put() for ever
            stop.set()
            worker.join(timeout=5.0)


def build_isic_dataset(image_dir, mask_dir, batch_size: int, image_size: int, augment: bool, shuffle: bool, seed: int):
    pairs = collect_isic_pairs(image_dir, mask_dir)
    return IsicDataset(pairs, batch_size, image_size, augment, shuffle, seed), len(pairs)


def prepare_isic_train_val_datasets(train_image_dir, train_mask_dir, val_image_dir, val_mask_dir, image_size: int,
                                    train_batch_size: int, val_batch_size: int, seed: int):
    """The official ISIC-2017 train / validation folders (:229-256): augmented + shuffled, and plain."""
    train_ds, n_train = build_isic_dataset(train_image_dir, train_mask_dir, train_batch_size, image_size, True, True, seed)
    val_ds, n_val = build_isic_dataset(val_image_dir, val_mask_dir, val_batch_size, image_size, False, False, seed)
    return train_ds, val_ds, n_train, n_val


# ----------------------------------------------------------------------------- training (:410-575)
def prepare_callbacks(run_dir: Path, ckpt_path: Path, patience):
    """Best-`val_dice` checkpoint, crash backup, per-epoch CSV (in place of TensorBoard), optional early stopping."""
    cbs = [ModelCheckpoint(ckpt_path, monitor="val_dice", mode="max", save_best_only=True),
           BackupAndRestore(run_dir / "train_backup"), CSVLogger(run_dir / "epoch_metrics.csv")]
    if patience is not None and patience > 0:
        cbs.append(EarlyStopping(monitor="val_dice", mode="max", patience=patience, restore_best_weights=True))
    return cbs


def train(args: argparse.Namespace):
    import torch
    protocol = PROTOCOLS[args.protocol]
    epochs = args.epochs or protocol.epochs
    batch_size = args.batch_size or protocol.batch_size
    for flag in ("train_images", "train_masks", "val_images", "val_masks"):
        if not getattr(args, flag):
            raise FileNotFoundError(f"--{flag} is required (the reference's cluster paths do not exist here)")
    folders = [Path(getattr(args, f)).expanduser() for f in ("train_images", "train_masks", "val_images", "val_masks")]
    train_ds, val_ds, n_train, n_val = prepare_isic_train_val_datasets(*folders, image_size=args.image_size,
                                                                       train_batch_size=batch_size, val_batch_size=batch_size,
                                                                       seed=args.seed)
    steps_per_epoch, val_steps = math.ceil(n_train / batch_size), math.ceil(n_val / batch_size)
    # --mixed_precision = mixed_float16 + dynamic loss scaling, as the reference (:471-476); --bf16 = this build's policy
    dtype = torch.bfloat16 if args.bf16 else torch.float16 if args.mixed_precision else torch.float32
    model = build_adaptive_depth_unet(args.image_size, args.base_channels, args.depth, dtype=dtype, seed=args.seed)
    model.compile(optimizer=build_optimizer(protocol, steps_per_epoch, epochs), loss=protocol.loss_builder(), jit_compile=False)
    summary: List[str] = []
    model.summary(print_fn=summary.append)
    model_dir, log_root = Path(args.model_dir).expanduser(), Path(args.log_dir).expanduser()
    stamp = datetime.now().strftime("%Y%m%d-%H%M%S")
    run_name = args.run_name or f"protocol{protocol.key}_seed{args.seed}_{stamp}"
    run_dir = log_root / run_name
    run_dir.mkdir(parents=True, exist_ok=True)
    model_dir.mkdir(parents=True, exist_ok=True)
    ckpt = model_dir / f"{run_name}.safetensors"
    patience = args.patience if args.patience is not None else protocol.early_stopping_patience
    history = model.fit(train_ds, epochs=epochs, validation_data=val_ds, callbacks=prepare_callbacks(run_dir, ckpt, patience),
                        verbose=1)
    metrics = model.evaluate(val_ds, return_dict=True)
    payload = {"protocol": protocol.key, "description": protocol.description, "epochs_requested": epochs,
               "epochs_ran": len(history.history.get("loss", [])), "initial_lr": protocol.initial_lr, "batch_size": batch_size,
               "image_size": args.image_size, "train_samples": n_train, "val_samples": n_val,
               "train_steps_per_epoch": steps_per_epoch, "val_steps": val_steps, "seed": args.seed,
               "mixed_precision": bool(args.mixed_precision), "threshold": DEFAULT_THRESHOLD, "model_checkpoint": str(ckpt),
               "train_images": str(folders[0]), "train_masks": str(folders[1]), "val_images": str(folders[2]),
               "val_masks": str(folders[3]), "metrics": metrics, "compute_dtype": str(dtype).replace("torch.", ""),
               "model_name": model.name}
    (run_dir / "config.json").write_text(json.dumps(payload, indent=2))
    (run_dir / "model_summary.txt").write_text("\n".join(summary))
    print("Validation metrics:")
    for key, value in metrics.items():
        print(f"  {key}: {value:.4f}")
    return history, metrics


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="Train Adaptive-Depth U-Net on ISIC-2017 segmentation.")
    p.add_argument("--protocol", type=str, choices=sorted(PROTOCOLS), default="A", help="Training protocol to follow.")
    p.add_argument("--epochs", type=int, default=0, help="Override epochs (0 keeps protocol default).")
    p.add_argument("--batch_size", type=int, default=0, help="Override batch size (0 keeps protocol default).")
    p.add_argument("--base_channels", type=int, default=DEFAULT_BASE_CHANNELS)
    p.add_argument("--depth", type=int, default=DEFAULT_DEPTH)
    p.add_argument("--image_size", type=int, default=DEFAULT_IMAGE_SIZE)
    p.add_argument("--seed", type=int, default=DEFAULT_SEED)
    p.add_argument("--patience", type=int, default=None, help="Override patience (None uses protocol default).")
    p.add_argument("--mixed_precision", action="store_true", help="Enable mixed_float16 policy.")
    p.add_argument("--bf16", action="store_true", help="bf16 activations (MI355X throughput policy; not a reference flag)")
    p.add_argument("--model_dir", type=str, default="models")
    p.add_argument("--log_dir", type=str, default="logs")
    p.add_argument("--run_name", type=str, default=None)
    p.add_argument("--train_images", type=str, default=None, help="Training image directory.")
    p.add_argument("--train_masks", type=str, default=None, help="Training mask directory.")
    p.add_argument("--val_images", type=str, default=None, help="Validation image directory.")
    p.add_argument("--val_masks", type=str, default=None, help="Validation mask directory.")
    return p.parse_args(argv)


if __name__ == "__main__":
    train(parse_args())
