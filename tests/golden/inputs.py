"""Seeded input generators shared by make_golden.py and the tests (data, not reference code)."""
import numpy as np
import torch


def coord_image(w, h):
    x = np.arange(w)[None, :].repeat(h, 0)
    y = np.arange(h)[:, None].repeat(w, 1)
    return np.stack([x % 256, y % 256, x // 256 + 16 * (y // 256)], -1).astype(np.uint8)


def seeded_boxes(n, seed, size=640.0):
    g = torch.Generator().manual_seed(seed)
    cx = torch.rand(n, generator=g) * size
    cy = torch.rand(n, generator=g) * size
    w = 16 + torch.rand(n, generator=g) * 240
    h = 16 + torch.rand(n, generator=g) * 240
    b = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).clamp(0, size)
    s = (torch.randperm(n, generator=g).float() + 0.25 + 0.5 * torch.rand(n, generator=g)) / n
    # no exact score ties (argsort tie order is unspecified in the reference)
    assert len(set(s.tolist())) == n
    return b, s
