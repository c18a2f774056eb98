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
