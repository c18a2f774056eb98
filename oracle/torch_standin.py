"""PyTorch-CPU stand-in for the reference's TF/Keras CPU path (test infrastructure and bench baseline only).

TensorFlow / Keras cannot be installed here or on the GPU box, so "the reference CPU path timed beside the
kernels"  is the same network rebuilt from a third party's CPU kernels:
F.conv2d (oneDNN) + F.layer_norm + F.interpolate(antialias=True) + autograd, float32, all host cores.  It follows
/root/reference/Super_resolution/code/train_adaptive_unet.py:200-287 (model), :316-320 (Charbonnier), :489-494 (Adam,
Keras epsilon placement) through the SRUNetOracle's layer list, and is cross-checked against the NumPy oracle in
tests/test_oracle_vs_torch.py.  It is a stand-in, clearly labelled as such wherever its numbers are printed: never a
parity source of truth by itself and never on the product path.
"""
from __future__ import annotations

import time
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .sr_unet import SRUNetOracle


class TorchSRUNet:
    def __init__(self, scale: float, depth: int, patch: int, base_channels: int = 64, head_channels: int = 64,
                 dtype=torch.float32):
        self.oracle = SRUNetOracle(scale, depth, patch, base_channels, head_channels)
        self.dtype = dtype
        self.P: Dict[str, torch.Tensor] = {}
        self.state: Dict[str, torch.Tensor] = {}
        self.step = 0

    def set_params(self, params: Dict[str, np.ndarray]):
        self.P = {k: torch.tensor(np.asarray(v), dtype=self.dtype, requires_grad=True) for k, v in params.items()}

    # ---- layers (NCHW tensors, Keras HWIO kernels)
    def _conv(self, x, name, k=3):
        return F.conv2d(x, self.P[name + "/kernel"].permute(3, 2, 0, 1), self.P[name + "/bias"], padding=k // 2)

    def _ln(self, x, name):
        y = F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), self.P[name + "/gamma"], self.P[name + "/beta"], eps=1e-3)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def _resize(x, s):
        return F.interpolate(x, size=(s, s), mode="bilinear", antialias=True, align_corners=False)

    def forward(self, lr_nhwc: torch.Tensor) -> torch.Tensor:
        m = self.oracle
        x = lr_nhwc.permute(0, 3, 1, 2)
        inp = x
        skips = []
        for step in m._plan:
            kind = step[0]
            if kind == "block":
                for conv, ln in step[1]:
                    x = F.relu(self._ln(self._conv(x, conv), ln))
            elif kind == "down":
                skips.append(x)
                x = self._resize(x, step[2])
            elif kind == "up":
                x = self._resize(x, step[2])
            elif kind == "upconv":
                x = F.relu(self._conv(x, step[1]))
            elif kind == "concat":
                x = torch.cat([x, skips[step[1]]], dim=1)
            elif kind == "head":
                x = torch.clamp(inp + self._conv(x, "residual_rgb", k=1), 0, 1)
        return x.permute(0, 2, 3, 1)

    def loss(self, lr_nhwc, hr_nhwc, eps: float = 1e-3):
        out = self.forward(lr_nhwc)
        return torch.sqrt((hr_nhwc - out) ** 2 + eps * eps).mean(), out

    def train_step(self, lr_nhwc, hr_nhwc, lr: float = 1e-4, b1=0.9, b2=0.999, eps=1e-7) -> float:
        """Forward, Charbonnier, autograd backward, Keras-form Adam (epsilon outside the bias correction)."""
        for p in self.P.values():
            p.grad = None
        loss, _ = self.loss(lr_nhwc, hr_nhwc)
        loss.backward()
        self.step += 1
        alpha = lr * (1 - b2 ** self.step) ** 0.5 / (1 - b1 ** self.step)
        with torch.no_grad():
            for k, p in self.P.items():
                m = self.state.setdefault("m/" + k, torch.zeros_like(p))
                v = self.state.setdefault("v/" + k, torch.zeros_like(p))
                m.mul_(b1).add_(p.grad, alpha=1 - b1)
                v.mul_(b2).addcmul_(p.grad, p.grad, value=1 - b2)
                p.addcdiv_(m, v.sqrt().add_(eps), value=-alpha)
        return float(loss.detach())


def time_train_steps(scale: float, depth: int, patch: int, batch: int, budget_seconds: float, seed: int = 1234):
    """Whole train steps of the stand-in on synthetic data (one untimed warm-up step, then steps until the budget is
    spent).  Returns (images_per_second, steps, seconds, threads)."""
    from .ops import cpu_share
    # one thread per CPU this process may actually use (its cgroup share: the GPU boxes show 256 logical CPUs to a process
    # whose quota is 16; 128 threads on that share ran the baseline at a fraction of what 16 do)
    torch.set_num_threads(cpu_share(cap=1 << 10))
    rng = np.random.default_rng(seed)
    net = TorchSRUNet(scale, depth, patch)
    net.set_params(net.oracle.init_params(rng, dtype=np.float32, head_uniform=0.05))
    hr = torch.from_numpy(rng.random((batch, patch, patch, 3), dtype=np.float32))
    lr = torch.clamp(hr + 0.05 * torch.from_numpy(rng.standard_normal(hr.shape, dtype=np.float32)), 0, 1)
    net.train_step(lr, hr)
    t0 = time.time()
    steps = 0
    while steps == 0 or time.time() - t0 < budget_seconds:
        net.train_step(lr, hr)
        steps += 1
    dt = time.time() - t0
    return batch * steps / dt, steps, dt, torch.get_num_threads()
