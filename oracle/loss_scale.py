"""Restatement of Keras' dynamic loss scaling (test infrastructure only).

tf.keras.mixed_precision.LossScaleOptimizer(inner, dynamic=True) is what `model.compile` wraps the optimizer in under
the reference's mixed_float16 policy (/root/reference/Super_resolution/code/train_adaptive_unet.py:471-477,
Segmenation/code/train_adaptive_unet.py:471-476).  Published behaviour (keras 3.3 / tf 2.16, the versions the reference
pins): initial_scale 2**15, dynamic_growth_steps 2000; each step the loss is multiplied by the scale before
differentiation and the gradients are divided by it; if any gradient is inf / NaN the update is skipped, the scale is
halved and the counter reset; otherwise the inner optimizer applies the update (its iteration count advances) and after
`dynamic_growth_steps` consecutive finite steps the scale doubles.  Parity unpinned against TensorFlow itself (not
installable here); the rules above are the documented contract.
"""
from __future__ import annotations

import numpy as np


class DynamicLossScale:
    def __init__(self, initial_scale: float = 2.0 ** 15, dynamic_growth_steps: int = 2000):
        self.scale = float(initial_scale)
        self.growth_steps = int(dynamic_growth_steps)
        self.good_steps = 0
        self.applied = 0
        self.skipped = 0

    def update(self, grads) -> bool:
        """grads: the SCALED gradients of this step.  Returns True when the optimizer may apply them (unscaled)."""
        finite = all(np.isfinite(g).all() for g in (grads.values() if isinstance(grads, dict) else grads))
        if not finite:
            self.scale = max(self.scale / 2.0, 1.0)
            self.good_steps = 0
            self.skipped += 1
            return False
        self.applied += 1
        self.good_steps += 1
        if self.good_steps >= self.growth_steps:
            self.scale *= 2.0
            self.good_steps = 0
        return True
