"""Host side of the training augmentation: draws, per training crop, the random numbers of
`data_transforms['train']` (utils/trainClass.py:199-216) and folds them into ONE record per sample that
`yv_augment_patchify` applies on the device in a single gather pass (csrc/augment.hip).

Reference sequence after Resize + Normalize (probabilities from the reference, parameter distributions from the
published albumentations 1.x transforms it names - the library is not in this image, so the distributions are
"parity unpinned"; what IS checked is that the device applies a given record exactly as the oracle states):

    HorizontalFlip p=.5 | [RandomCrop(200,200) + PadIfNeeded(S,S)] p=.25 | ShiftScaleRotate(shift .0625, scale .05,
    rotate 10) p=.25 | ChannelShuffle p=.5 | OneOf[GridDistortion(5, .05), ElasticTransform(1, 50, 50)] p=.25 |
    CoarseDropout(5..8 holes of S//20, fill 0) p=.5

Deliberate differences, both confined to samples where two resampling transforms fire together (p = 1/16):
the continuous maps are composed and the image is resampled ONCE (bilinear, BORDER_REFLECT_101) instead of once per
transform; and ElasticTransform's Gaussian displacement field (alpha 1 blurred with sigma 50: < 0.02 px) is dropped,
its random affine (alpha_affine 50) is kept.  PadIfNeeded is given `value=[0,0,0]` without a border mode by the
reference, so the library's default BORDER_REFLECT_101 applies and the value is unused; this build does the same.
"""
from __future__ import annotations

import math
import random
from typing import Optional, Tuple

import numpy as np

GEO_HEAD, IDX_HEAD, MAX_HOLES = 6, 36, 8
CROP = 200                                            # RandomCrop(height=200, width=200), utils/trainClass.py:204


def reflect101(i: np.ndarray, n: int) -> np.ndarray:
    """cv2.BORDER_REFLECT_101 index: gfedcb|abcdefgh|gfedcba."""
    if n == 1:
        return np.zeros_like(i)
    per = 2 * n - 2
    m = np.mod(i, per)
    return np.where(m < n, m, per - m)


def index_tables(S: int, flip: bool, crop_xy: Optional[Tuple[int, int]], crop: int = CROP):
    """Integer column / row tables of the flipped, crop-padded frame J: J[y][x] = I[mapy[y]][mapx[x]]."""
    ax = np.arange(S, dtype=np.int64)
    mx, my = ax.copy(), ax.copy()
    if crop_xy is not None and crop <= S:
        x1, y1 = crop_xy
        left = int((S - crop) / 2.0)                  # PadIfNeeded position=center
        mx = x1 + reflect101(ax - left, crop)
        my = y1 + reflect101(ax - left, crop)
    if flip:
        mx = S - 1 - mx
    return mx.astype(np.int32), my.astype(np.int32)


def ssr_matrix(S: int, angle: float, scale: float, dx: float, dy: float) -> np.ndarray:
    """Forward 3x3 matrix of ShiftScaleRotate: cv2.getRotationMatrix2D((S/2-.5, S/2-.5), angle, scale) with the
    translation (dx*S, dy*S) added; warpAffine samples the source at its inverse."""
    c = S / 2.0 - 0.5
    a, b = scale * math.cos(math.radians(angle)), scale * math.sin(math.radians(angle))
    return np.array([[a, b, (1 - a) * c - b * c + dx * S], [-b, a, b * c + (1 - a) * c + dy * S], [0, 0, 1]], dtype=np.float64)


def elastic_matrix(S: int, jitter: np.ndarray) -> np.ndarray:
    """Forward 3x3 matrix of ElasticTransform's random affine: cv2.getAffineTransform(pts1, pts1 + jitter) with
    pts1 the three corners around the centre at distance S//3; jitter (3,2) ~ U(-alpha_affine, alpha_affine)."""
    c, q = float(S // 2), float(S // 3)
    p1 = np.array([[c + q, c + q], [c + q, c - q], [c - q, c - q]], dtype=np.float64)
    p2 = p1 + np.asarray(jitter, dtype=np.float64)
    A = np.concatenate([p1, np.ones((3, 1))], axis=1)               # A @ M^T = p2
    M = np.linalg.solve(A, p2).T
    return np.concatenate([M, [[0.0, 0.0, 1.0]]], axis=0)


def grid_lut(S: int, steps: np.ndarray, num_steps: int = 5) -> np.ndarray:
    """Per-axis source coordinate table of GridDistortion: cells of S//num_steps pixels, cell i stretched by steps[i];
    the last (partial) cell ends at S."""
    cell = S // num_steps
    out = np.zeros(S, dtype=np.float64)
    prev = 0.0
    for i in range(num_steps + 1):
        start, end = i * cell, i * cell + cell
        if end > S:
            end, cur = S, float(S)
        else:
            cur = prev + cell * float(steps[i])
        if end > start:
            out[start:end] = np.linspace(prev, cur, end - start)
        prev = cur
    return out


def identity_record(S: int):
    geo = np.zeros(GEO_HEAD + 2 * S, dtype=np.float32)
    geo[[0, 4]] = 1.0
    geo[GEO_HEAD:GEO_HEAD + S] = np.arange(S)
    geo[GEO_HEAD + S:] = np.arange(S)
    idx = np.zeros(IDX_HEAD + 2 * S, dtype=np.int32)
    idx[0:3] = (0, 1, 2)
    idx[IDX_HEAD:IDX_HEAD + S] = np.arange(S)
    idx[IDX_HEAD + S:] = np.arange(S)
    return geo, idx


def make_record(S: int, flip=False, crop_xy=None, ssr=None, perm=(0, 1, 2), grid=None, elastic=None, holes=()):
    """Fold one sample's drawn parameters into (geo, idx).  ssr = (angle, scale, dx, dy); grid = (stepsx, stepsy) with
    num_steps+1 entries each; elastic = (3,2) jitter; holes = [(x1,y1,x2,y2)], at most 8."""
    geo, idx = identity_record(S)
    inv = np.eye(3)
    if ssr is not None:
        inv = np.linalg.inv(ssr_matrix(S, *ssr))
    if elastic is not None:
        inv = inv @ np.linalg.inv(elastic_matrix(S, elastic))
    geo[0:6] = inv[:2].reshape(-1).astype(np.float32)
    if grid is not None:
        geo[GEO_HEAD:GEO_HEAD + S] = grid_lut(S, grid[0], len(grid[0]) - 1).astype(np.float32)
        geo[GEO_HEAD + S:] = grid_lut(S, grid[1], len(grid[1]) - 1).astype(np.float32)
    idx[0:3] = perm
    holes = list(holes)[:MAX_HOLES]
    idx[3] = len(holes)
    for h, q in enumerate(holes):
        idx[4 + 4 * h:8 + 4 * h] = q
    mx, my = index_tables(S, flip, crop_xy)
    idx[IDX_HEAD:IDX_HEAD + S], idx[IDX_HEAD + S:] = mx, my
    return geo, idx


class TrainAugment:
    """Draws the records of `data_transforms['train']`.  Seeded from Python's `random` unless a seed is given, so
    `set_seed()` (utils/trainClass.py:330-337) makes a run repeatable."""

    def __init__(self, size: int, seed: Optional[int] = None):
        self.S = int(size)
        self.rng = np.random.default_rng(random.getrandbits(63) if seed is None else seed)

    def draw(self) -> dict:
        S, r = self.S, self.rng
        p = dict(flip=bool(r.random() < 0.5), crop_xy=None, ssr=None, perm=(0, 1, 2), grid=None, elastic=None, holes=())
        if r.random() < 0.25 and CROP <= S:
            hs, ws = r.random(), r.random()                        # RandomCrop: int((S - crop + 1) * u)
            p["crop_xy"] = (int((S - CROP + 1) * ws), int((S - CROP + 1) * hs))
        if r.random() < 0.25:
            p["ssr"] = (r.uniform(-10, 10), r.uniform(0.95, 1.05), r.uniform(-0.0625, 0.0625), r.uniform(-0.0625, 0.0625))
        if r.random() < 0.5:
            p["perm"] = tuple(int(v) for v in r.permutation(3))
        if r.random() < 0.25:
            if r.random() < 0.5:
                p["grid"] = (1 + r.uniform(-0.05, 0.05, 6), 1 + r.uniform(-0.05, 0.05, 6))
            else:
                p["elastic"] = r.uniform(-50, 50, (3, 2))
        if r.random() < 0.5:
            hole = max(S // 20, 1)                                  # max_height = img_size // 20; min_* default to max_*
            holes = []
            for _ in range(int(r.integers(5, MAX_HOLES + 1))):
                y1, x1 = int(r.integers(0, S - hole + 1)), int(r.integers(0, S - hole + 1))
                holes.append((x1, y1, x1 + hole, y1 + hole))
            p["holes"] = holes
        return p

    def sample(self, B: int):
        """-> geo (B, 6+2S) f32, idx (B, 36+2S) i32 host arrays for `yvhip.augment_patchify`."""
        recs = [make_record(self.S, **self.draw()) for _ in range(B)]
        return np.stack([g for g, _ in recs]), np.stack([i for _, i in recs])
