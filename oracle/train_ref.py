"""TEST INFRASTRUCTURE ONLY -- CPU restatement of one training step of the reference driver.

Follows ``CNN implementation/multimodal_train.py:175-202``: zero grads, forward with labels,
backward, ``clip_grad_norm_(max_norm=1.0)``, ``AdamW.step``; optimizer settings from ``:481-485``
(torch defaults: betas (0.9, 0.999), eps 1e-8, decoupled weight decay).  The clip and AdamW
arithmetic is written out explicitly (not through ``torch.optim``) so it can be compared term by
term with the fused HIP kernels; ``tests/test_oracle_golden.py`` pins it against a loss curve
produced by the reference model driven by stock ``torch.optim.AdamW``.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from . import model_ref


class TrainState:
    def __init__(self, sd: Dict[str, torch.Tensor], modality: str, lr: float = 1e-3, weight_decay: float = 1e-4,
                 betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 1.0):
        self.sd = model_ref.clone_state(sd)
        self.modality = modality
        self.lr, self.wd, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_norm
        self.keys = model_ref.param_keys(self.sd)
        self.m = {k: torch.zeros_like(self.sd[k]) for k in self.keys}
        self.v = {k: torch.zeros_like(self.sd[k]) for k in self.keys}
        self.t = 0

    def step(self, audio, input_ids, labels, drop_audio=False, drop_text=False):
        out, grads, bufs = model_ref.forward_backward(self.sd, self.modality, audio, input_ids, labels,
                                                      drop_audio, drop_text)
        for k, v in bufs.items():          # BatchNorm running statistics advance with the forward
            self.sd[k] = v
        # clip_grad_norm_: norm of per-tensor norms, coefficient clamped to 1 (multimodal_train.py:194)
        total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(grads[k]) for k in self.keys]))
        coef = torch.clamp(self.max_norm / (total + 1e-6), max=1.0)
        self.t += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.t
        bc2_sqrt = math.sqrt(1.0 - b2 ** self.t)
        step_size = self.lr / bc1
        for k in self.keys:
            g = grads[k] * coef
            p = self.sd[k]
            p.mul_(1.0 - self.lr * self.wd)
            self.m[k].lerp_(g, 1.0 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1.0 - b2)
            denom = (self.v[k].sqrt() / bc2_sqrt).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-step_size)
        return {"loss": float(out["loss"]), "logits": out["logits"], "grad_norm": float(total)}
