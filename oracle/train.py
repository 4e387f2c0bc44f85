"""Oracle (TEST INFRASTRUCTURE): loss, schedule, accuracy, optimizer step of
the ViT fine-tune rows C1-C3 of SURVEY.md section 8(a).  All pinned by golden
vectors captured from the reference's own functions (tests/golden)."""
from __future__ import annotations

import math

import numpy as np
import torch


def lsce(x: torch.Tensor, y_onehot: torch.Tensor, smoothing: float = 0.1) -> torch.Tensor:
    """utils/trainClass.py:162-185: log(softmax) form (not log_softmax)."""
    target = torch.max(y_onehot, 1).indices
    y_hat = torch.softmax(x, dim=1)
    cross = -torch.log(y_hat[torch.arange(len(y_hat)), target])
    smooth = -torch.log(y_hat).mean(dim=1)
    return ((1.0 - smoothing) * cross + smoothing * smooth).mean()


def focal(x: torch.Tensor, y_onehot: torch.Tensor, alpha: float = 1.0, gamma: float = 2.0) -> torch.Tensor:
    """utils/trainClass.py:46-66 (reduction='mean' over B*C elements)."""
    bce = torch.nn.functional.binary_cross_entropy_with_logits(x, y_onehot, reduction="none")
    p_t = torch.exp(-bce)
    return (alpha * (1 - p_t) ** gamma * bce).mean()


def build_loss(x: torch.Tensor, y_onehot: torch.Tensor) -> torch.Tensor:
    """utils/trainClass.py:362-370: LSCE(0.1)/6 + Focal*5/6."""
    return lsce(x, y_onehot) / 6 + focal(x, y_onehot) * 5 / 6


def cosine_lr(t: int, nb_epoch: int, lr: float) -> float:
    """utils/trainClass.py:97-105 (numpy double arithmetic)."""
    cos_inner = np.pi * (t % nb_epoch)
    cos_inner /= nb_epoch
    return float(lr / 2 * (np.cos(cos_inner) + 1))


def get_correct(output: torch.Tensor, target_onehot: torch.Tensor, num_classes: int = 5):
    """utils/trainClass.py:109-117 without sklearn: eq vector + confusion
    matrix [true][pred] over labels range(num_classes)."""
    pred = torch.max(output, 1).indices
    tgt = torch.max(target_onehot, 1).indices
    cm = np.zeros((num_classes, num_classes), dtype=np.int64)
    for t, p in zip(tgt.tolist(), pred.tolist()):
        cm[t, p] += 1
    return pred.eq(tgt), cm


def sgd_step(p: torch.Tensor, g: torch.Tensor, buf, lr: float, momentum: float = 0.9,
             weight_decay: float = 1e-3):
    """torch.optim.SGD as configured at utils/trainClass.py:442-443
    (dampening 0, no nesterov): g += wd*p; buf = g (first) | mu*buf + g;
    p -= lr*buf.  Returns (p_new, buf_new)."""
    g = g + weight_decay * p
    buf = g.clone() if buf is None else momentum * buf + g
    return p - lr * buf, buf
