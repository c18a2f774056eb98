#!/usr/bin/env python3
"""Vanilla segmentation baseline's own entry point: mirror of /root/reference/Segmenation/code/unet_vinillia.py:101-297.

`build_unet` (LayerNorm conv blocks, MaxPooling2D, Conv2DTranspose(nf, 2, strides=2) decoder, sigmoid head "mask_logits")
lives in seg_model.py; this file is the reference script's data path, metric set and callbacks around it:

* `_canonical_key` / `_discover_pairs` (:102-155): recursive listing by suffix, natural sort, image <-> mask matching on the
  file stem with the dataset's decoration tokens removed; same errors (`ValueError`);
* `_parse_example` (:158-174): image decoded to RGB, resized to `image_size` with tf.image.resize's BILINEAR (no antialias,
  half-pixel centres), / 255; mask resized with NEAREST_NEIGHBOR (half-pixel centres), / 255, thresholded at 0.5;
* `_augment` (:177-184): independent left-right / up-down flips with probability 1/2;
* `build_dataset` (:187-207): shuffle over the whole set (reshuffled every pass), batches, last partial batch kept;
* `train` (:236-293): BinaryCrossentropy loss, metrics BinaryAccuracy / Precision / Recall / dice_coefficient, Adam(lr),
  ModelCheckpoint + EarlyStopping on `val_dice_coefficient` (max), ReduceLROnPlateau(val_loss, 0.5, 5, min_lr 1e-6),
  `<run_name>_best` / `<run_name>_final` checkpoints (flat .safetensors here: Keras' own archive format is f3's open part).

TensorFlow's JPEG decoder and resize kernels are not available here: pixel values are parity-unpinned (Pillow decodes; the
two resize rules are restated from tf.image.resize's documented half-pixel-centre definitions and tested on closed forms).
"""
from __future__ import annotations

import argparse
from pathlib import Path
from typing import Iterator, List, Sequence, Tuple

import numpy as np

from .callbacks import EarlyStopping, ModelCheckpoint, ReduceLROnPlateau
from .pipeline import sorted_alphanumeric
from .seg_model import Adam, binary_crossentropy, build_unet

DEFAULT_IMAGE_SUFFIX = ".jpg"
DEFAULT_MASK_SUFFIX = "_segmentation.png"
METRICS = ["accuracy", "precision", "recall", "dice_coefficient"]          # unet_vinillia.py:266-271


def dice_coefficient(y_true: np.ndarray, y_pred: np.ndarray, smooth: float = 1e-6) -> float:
    """unet_vinillia.py:94-99 on host arrays (the train / eval loops take it from the head kernel's sums)."""
    y_true, y_pred = np.asarray(y_true, np.float32), np.asarray(y_pred, np.float32)
    return float((2.0 * np.sum(y_true * y_pred) + smooth) / (np.sum(y_true + y_pred) + smooth))


def _canonical_key(path: Path) -> str:
    stem = path.stem.lower()
    for token in ("_segmentation", "_mask", "_leftimg8bit", "_gtfine_labelids", "_gtfine_polygons", "_gtfine_color",
                  "_gtfine_instanceids", "_gtcoarse_labelids", "_gtcoarse_color", "_gtcoarse_instanceids", "_instanceids"):
        stem = stem.replace(token, "")
    return stem


def _discover_pairs(image_dir: Path, mask_dir: Path, image_suffix: str, mask_suffix: str, limit) -> List[Tuple[str, str]]:
    image_paths = [Path(p) for p in sorted_alphanumeric([str(p) for p in image_dir.rglob(f"*{image_suffix}") if p.is_file()])]
    mask_lookup = {_canonical_key(p): p for p in mask_dir.rglob(f"*{mask_suffix}") if p.is_file()}
    if not image_paths:
        raise ValueError(f"No images found in {image_dir} with suffix {image_suffix}")
    if not mask_lookup:
        raise ValueError(f"No masks found in {mask_dir} with suffix {mask_suffix}")
    pairs = []
    for image_path in image_paths:
        key = _canonical_key(image_path)
        mask_path = mask_lookup.get(key)
        if mask_path is None:
            raise ValueError(f"Missing mask for image {image_path.name} (expected key {key})")
        pairs.append((str(image_path), str(mask_path)))
    return pairs[:limit] if limit is not None else pairs


def resize_bilinear(img: np.ndarray, size: int) -> np.ndarray:
    """tf.image.resize(..., BILINEAR) without antialias: output pixel i samples the input at (i + 0.5) * in / out - 0.5,
    clamped to the image, linear between the two neighbours; separable, float32."""
    img = np.asarray(img, np.float32)

    def axis_tables(n_in):
        pos = (np.arange(size, dtype=np.float32) + np.float32(0.5)) * np.float32(n_in / size) - np.float32(0.5)
        lo = np.floor(pos)
        frac = (pos - lo).astype(np.float32)
        i0 = np.clip(lo.astype(np.int64), 0, n_in - 1)
        i1 = np.clip(lo.astype(np.int64) + 1, 0, n_in - 1)
        return i0, i1, frac

    y0, y1, fy = axis_tables(img.shape[0])
    x0, x1, fx = axis_tables(img.shape[1])
    rows = img[y0] + (img[y1] - img[y0]) * fy[:, None, None]
    return (rows[:, x0] + (rows[:, x1] - rows[:, x0]) * fx[None, :, None]).astype(np.float32)


def resize_nearest(img: np.ndarray, size: int) -> np.ndarray:
    """tf.image.resize(..., NEAREST_NEIGHBOR): output pixel i takes input floor((i + 0.5) * in / out) (half-pixel centres)."""
    def idx(n_in):
        return np.minimum(np.floor((np.arange(size, dtype=np.float32) + np.float32(0.5)) * np.float32(n_in / size)).astype(np.int64),
                          n_in - 1)
    return img[idx(img.shape[0])][:, idx(img.shape[1])]


def _parse_example(image_path: str, mask_path: str, image_size: int) -> Tuple[np.ndarray, np.ndarray]:
    from PIL import Image
    with Image.open(image_path) as im:
        image = np.asarray(im.convert("RGB"), np.float32)
    image = resize_bilinear(image, image_size) / np.float32(255.0)
    with Image.open(mask_path) as im:
        mask = np.asarray(im.convert("L"), np.float32)[..., None]
    mask = resize_nearest(mask, image_size) / np.float32(255.0)
    return image, np.where(mask > 0.5, 1.0, 0.0).astype(np.float32)


def _augment(image: np.ndarray, mask: np.ndarray, rng: np.random.Generator) -> Tuple[np.ndarray, np.ndarray]:
    if rng.random() > 0.5:
        image, mask = image[:, ::-1], mask[:, ::-1]
    if rng.random() > 0.5:
        image, mask = image[::-1], mask[::-1]
    return np.ascontiguousarray(image), np.ascontiguousarray(mask)


class PairDataset:
    """build_dataset (:187-207) as a re-iterable of (images [B,S,S,3], masks [B,S,S,1]) float32 batches."""

    def __init__(self, pairs: Sequence[Tuple[str, str]], image_size: int, batch_size: int, shuffle: bool, augment: bool, seed: int):
        if batch_size <= 0:
            raise ValueError("batch_size must be positive")
        self.pairs, self.size, self.batch_size = list(pairs), image_size, batch_size
        self.shuffle, self.augment, self.seed = shuffle, augment, seed
        self._pass = 0

    def __len__(self) -> int:
        return (len(self.pairs) + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Tuple[np.ndarray, np.ndarray]]:
        rng = np.random.default_rng(self.seed + self._pass)               # reshuffle_each_iteration=True
        self._pass += 1
        order = rng.permutation(len(self.pairs)) if self.shuffle else np.arange(len(self.pairs))
        imgs, masks = [], []
        for i in order:
            img, msk = _parse_example(*self.pairs[int(i)], self.size)
            if self.augment:
                img, msk = _augment(img, msk, rng)
            imgs.append(img)
            masks.append(msk)
            if len(imgs) == self.batch_size:
                yield np.stack(imgs), np.stack(masks)
                imgs, masks = [], []
        if imgs:
            yield np.stack(imgs), np.stack(masks)


def build_dataset(pairs, image_size: int, batch_size: int, shuffle: bool, augment: bool, seed: int) -> PairDataset:
    return PairDataset(pairs, image_size, batch_size, shuffle, augment, seed)


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="Train a baseline U-Net on the ISIC-2017 dataset.")
    p.add_argument("--train_image_dir", type=Path, default=None, help="Directory of training images.")
    p.add_argument("--train_mask_dir", type=Path, default=None, help="Directory of training segmentation masks.")
    p.add_argument("--val_image_dir", type=Path, default=None, help="Directory of validation images.")
    p.add_argument("--val_mask_dir", type=Path, default=None, help="Directory of validation masks.")
    p.add_argument("--image_suffix", type=str, default=DEFAULT_IMAGE_SUFFIX, help="Suffix/pattern for image files.")
    p.add_argument("--mask_suffix", type=str, default=DEFAULT_MASK_SUFFIX, help="Suffix/pattern for mask files.")
    p.add_argument("--image_size", type=int, default=256, help="Square input resolution.")
    p.add_argument("--batch_size", type=int, default=8, help="Batch size.")
    p.add_argument("--epochs", type=int, default=60, help="Number of training epochs.")
    p.add_argument("--learning_rate", type=float, default=1e-4, help="Adam learning rate.")
    p.add_argument("--base_channels", type=int, default=32, help="Number of filters in the first encoder block.")
    p.add_argument("--depth", type=int, default=4, help="Depth of the encoder/decoder.")
    p.add_argument("--model_dir", type=Path, default=Path("models"), help="Directory to save checkpoints.")
    p.add_argument("--run_name", type=str, default="unet_isic", help="Prefix for saved checkpoints.")
    p.add_argument("--seed", type=int, default=13, help="Random seed for shuffling.")
    p.add_argument("--limit_train", type=int, default=None, help="Optional limit on number of training samples.")
    p.add_argument("--limit_val", type=int, default=None, help="Optional limit on number of validation samples.")
    p.add_argument("--augment", action="store_true", help="Enable simple geometric augmentations.")
    p.add_argument("--mixed_precision", action="store_true", help="Use the mixed_float16 policy (fp16 kernels + dynamic loss scaling).")
    p.add_argument("--dtype", choices=["float32", "bfloat16", "float16"], default=None,
                   help="compute dtype (default float32, or float16 with --mixed_precision; bfloat16 is this build's throughput type)")
    p.add_argument("--fit_verbose", type=int, choices=[0, 1, 2], default=2, help="Keras verbosity mode.")
    return p.parse_args(argv)


def train(args: argparse.Namespace):
    import torch
    dirs = []
    for attr, label in (("train_image_dir", "training images"), ("train_mask_dir", "training masks"),
                        ("val_image_dir", "validation images"), ("val_mask_dir", "validation masks")):
        d = getattr(args, attr)
        if d is None or not Path(d).expanduser().exists():
            raise FileNotFoundError(f"Missing {label} directory: {d}")
        dirs.append(Path(d).expanduser())
    train_pairs = _discover_pairs(dirs[0], dirs[1], args.image_suffix, args.mask_suffix, args.limit_train)
    val_pairs = _discover_pairs(dirs[2], dirs[3], args.image_suffix, args.mask_suffix, args.limit_val)
    train_ds = build_dataset(train_pairs, args.image_size, args.batch_size, shuffle=True, augment=args.augment, seed=args.seed)
    val_ds = build_dataset(val_pairs, args.image_size, args.batch_size, shuffle=False, augment=False, seed=args.seed)
    print(f"Loaded {len(train_pairs)} training samples and {len(val_pairs)} validation samples.")

    dtype = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16,
             None: torch.float16 if args.mixed_precision else torch.float32}[args.dtype]
    model = build_unet(args.image_size, num_classes=1, base_channels=args.base_channels, depth=args.depth, dtype=dtype)
    model.compile(optimizer=Adam(learning_rate=args.learning_rate), loss=binary_crossentropy(), metrics=METRICS)

    model_dir = Path(args.model_dir).expanduser()
    model_dir.mkdir(parents=True, exist_ok=True)
    checkpoint_path = model_dir / f"{args.run_name}_best.safetensors"
    print(f"Checkpoints will be written to {checkpoint_path}")
    callbacks = [ModelCheckpoint(checkpoint_path, monitor="val_dice_coefficient", mode="max", save_best_only=True),
                 EarlyStopping(monitor="val_dice_coefficient", patience=10, mode="max", restore_best_weights=True),
                 ReduceLROnPlateau(monitor="val_loss", factor=0.5, patience=5, min_lr=1e-6, verbose=1)]
    history = model.fit(train_ds, validation_data=val_ds, epochs=args.epochs, callbacks=callbacks, verbose=args.fit_verbose)
    final_path = model_dir / f"{args.run_name}_final.safetensors"
    model.save_weights(final_path)
    return model, history


if __name__ == "__main__":
    train(parse_args())
