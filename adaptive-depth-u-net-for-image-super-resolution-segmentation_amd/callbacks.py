"""Keras-callback-shaped helpers used by train() (train_adaptive_unet.py:600-620)."""
from __future__ import annotations

import math
from pathlib import Path


class Callback:
    def set_model(self, model):
        self.model = model


class EarlyStopping(Callback):
    """EarlyStopping(monitor="val_loss", patience, restore_best_weights=True)."""

    def __init__(self, monitor="val_loss", patience=10, restore_best_weights=True, mode="min"):
        self.monitor, self.patience, self.restore, self.mode = monitor, patience, restore_best_weights, mode
        self.best, self.wait, self.best_weights = None, 0, None

    def _better(self, v):
        return self.best is None or (v < self.best if self.mode == "min" else v > self.best)

    def on_epoch_end(self, epoch, logs):
        v = logs.get(self.monitor)
        if v is None or math.isnan(v):
            return
        if self._better(v):
            self.best, self.wait = v, 0
            if self.restore:      # trainable parameters AND the non-trainable state (BatchNorm moving statistics)
                self.best_weights = (self.model.P.clone(), [t.clone() for t in self.model._graph_extra_state()])
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.model.stop_training = True
                if self.restore and self.best_weights is not None:
                    self.model.P.copy_(self.best_weights[0])
                    for t, saved in zip(self.model._graph_extra_state(), self.best_weights[1]):
                        t.copy_(saved)
                    self.model._repack()


class ReduceLROnPlateau(Callback):
    """tf.keras.callbacks.ReduceLROnPlateau(monitor="val_loss", factor, patience, min_lr) as the vanilla segmentation baseline
    uses it (Segmenation/code/unet_vinillia.py:283): when `monitor` has not improved by more than `min_delta` for `patience`
    epochs the optimizer's learning rate is multiplied by `factor` (not below `min_lr`), then `cooldown` epochs pass before
    the count resumes.  The optimizer must hold a plain float learning rate (Keras raises for a schedule as well); the new
    value reaches the kernels with the next step's step-size publication, also under graph replay."""

    def __init__(self, monitor="val_loss", factor=0.1, patience=10, verbose=0, mode="auto", min_delta=1e-4, cooldown=0, min_lr=0.0):
        if factor >= 1.0:
            raise ValueError("ReduceLROnPlateau does not support a factor >= 1.0.")
        if mode not in ("auto", "min", "max"):
            mode = "auto"
        self.monitor, self.factor, self.patience, self.verbose = monitor, factor, patience, verbose
        self.min_delta, self.cooldown, self.min_lr = min_delta, cooldown, min_lr
        self.mode = ("max" if "acc" in monitor else "min") if mode == "auto" else mode
        self.on_train_begin({})

    def on_train_begin(self, logs):
        self.best = math.inf if self.mode == "min" else -math.inf
        self.wait = self.cooldown_counter = 0

    def _improved(self, v):
        return v < self.best - self.min_delta if self.mode == "min" else v > self.best + self.min_delta

    def on_epoch_end(self, epoch, logs):
        opt = self.model.optimizer
        inner = getattr(opt, "inner_optimizer", opt)                   # LossScaleOptimizer wraps the Adam that holds the rate
        if callable(inner.learning_rate):
            raise TypeError("ReduceLROnPlateau needs a float learning rate, not a schedule")
        logs["learning_rate"] = float(inner.learning_rate)
        v = logs.get(self.monitor)
        if v is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self._improved(v):
            self.best, self.wait = v, 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                old = float(inner.learning_rate)
                if old > self.min_lr:
                    inner.learning_rate = max(old * self.factor, self.min_lr)
                    if self.verbose:
                        print(f"\nEpoch {epoch + 1}: ReduceLROnPlateau reducing learning rate to {inner.learning_rate}.")
                    self.cooldown_counter = self.cooldown
                    self.wait = 0


class ModelCheckpoint(Callback):
    """ModelCheckpoint(filepath, monitor="val_loss", save_best_only=True) writing flat .safetensors files."""

    def __init__(self, filepath, monitor="val_loss", save_best_only=True, mode="min"):
        self.filepath, self.monitor, self.best_only, self.mode = Path(filepath), monitor, save_best_only, mode
        self.best = None

    def on_epoch_end(self, epoch, logs):
        v = logs.get(self.monitor)
        better = v is not None and (self.best is None or (v < self.best if self.mode == "min" else v > self.best))
        if better or not self.best_only:
            if better:
                self.best = v
            self.filepath.parent.mkdir(parents=True, exist_ok=True)
            self.model.save_weights(self.filepath)


class CSVLogger(Callback):
    """Writes epoch_metrics.csv with the columns of export_log_metrics.py:109-119."""

    def __init__(self, path):
        self.path = Path(path)
        self.rows = []

    def on_epoch_end(self, epoch, logs):
        self.rows.append({"epoch": epoch + 1, **logs})
        self.path.parent.mkdir(parents=True, exist_ok=True)
        keys = ["epoch"] + sorted({k for r in self.rows for k in r if k != "epoch"})
        with self.path.open("w") as fh:
            fh.write(",".join(keys) + "\n")
            for r in self.rows:
                fh.write(",".join(str(r.get(k, "")) for k in keys) + "\n")


class BackupAndRestore(Callback):
    """tf.keras.callbacks.BackupAndRestore(backup_dir) (train_adaptive_unet.py:615): after every epoch the whole training
    state (weights, non-trainable statistics, Adam moments, iteration count, loss scaler, epoch index) goes to
    `backup_dir`; a later fit() with the same directory continues after the last finished epoch; the backup is deleted
    when training ends normally.  The dataset position is not part of the state (as in Keras)."""

    def __init__(self, backup_dir):
        self.dir = Path(backup_dir)
        self.file = self.dir / "backup.safetensors"
        self.meta = self.dir / "backup.json"

    def restore_training_state(self, model) -> int:
        """Called by fit() before the first epoch; returns the epoch to start from (0 without a backup)."""
        import json
        if not (self.file.exists() and self.meta.exists()):
            return 0
        model.load_weights(self.file, restore_optimizer=True)
        return int(json.loads(self.meta.read_text())["epochs_done"])

    def on_epoch_end(self, epoch, logs):
        import json
        self.dir.mkdir(parents=True, exist_ok=True)
        tmp = self.dir / "backup.tmp.safetensors"
        self.model.save_weights(tmp, include_optimizer=True)
        tmp.replace(self.file)                                   # atomic: a crash mid-write keeps the previous backup
        self.meta.write_text(json.dumps({"epochs_done": epoch + 1}))

    def on_train_end(self, logs):
        for f in (self.file, self.meta):
            if f.exists():
                f.unlink()
