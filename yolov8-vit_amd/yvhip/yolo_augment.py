"""Host side of the detector's training augmentation: what `model.train(...)` of
utils/trainYolo.py:28 applies with the ultralytics defaults - Mosaic(p 1.0, four images around a random centre of a
2S canvas) -> RandomPerspective(degrees 0, translate 0.1, scale 0.5, shear 0) -> RandomHSV(0.015, 0.7, 0.4) ->
RandomFlip(lr 0.5); without mosaic (`close_mosaic` epochs): LetterBox -> the same affine on the S canvas.

The host draws the random numbers, decodes the files and transforms the LABELS (a few boxes per image); every pixel
is produced on the device: sources are resized by `yv_letterbox` into S x S tiles and `yv_mosaic_augment` gathers each
output image through the inverse affine straight from the tiles (the 2S x 2S canvas is never materialised).  Parity
unpinned: the pipeline lives in `ultralytics` / OpenCV (absent); 8-bit HSV and the bilinear rounding are this build's
own statements (oracle/yolo_augment.py), which can differ from OpenCV's fixed-point paths by one grey level.
"""
from __future__ import annotations

import math
import os
import random
from typing import List, Optional, Sequence, Tuple

import numpy as np

FILL = 114


def tile_geometry(w0: int, h0: int, S: int) -> Tuple[int, int]:
    """Size of a source after the loader's resize: long side -> S, ceil, capped at S."""
    r = S / max(h0, w0)
    if r == 1:
        return w0, h0
    return min(math.ceil(w0 * r), S), min(math.ceil(h0 * r), S)


def mosaic_placement(i: int, xc: int, yc: int, w: int, h: int, S: int):
    """Canvas rectangle (x1a,y1a,x2a,y2a) of quadrant i (0 top-left, 1 top-right, 2 bottom-left, 3 bottom-right) around
    the centre (xc,yc) of the 2S canvas and the tile pixel (x1b,y1b) shown at its top-left corner."""
    s2 = 2 * S
    if i == 0:
        x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
        x1b, y1b = w - (x2a - x1a), h - (y2a - y1a)
    elif i == 1:
        x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, s2), yc
        x1b, y1b = 0, h - (y2a - y1a)
    elif i == 2:
        x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(s2, yc + h)
        x1b, y1b = w - (x2a - x1a), 0
    else:
        x1a, y1a, x2a, y2a = xc, yc, min(xc + w, s2), min(s2, yc + h)
        x1b, y1b = 0, 0
    return (x1a, y1a, x2a, y2a), (x1b, y1b)


def affine_matrix(canvas: int, S: int, scale: float, tx: float, ty: float) -> np.ndarray:
    """Forward 3x3 matrix T @ R @ C of RandomPerspective with degrees = shear = perspective = 0: centre the canvas,
    scale, translate by (tx, ty) * S."""
    C = np.array([[1, 0, -canvas / 2], [0, 1, -canvas / 2], [0, 0, 1]], dtype=np.float64)
    R = np.diag([scale, scale, 1.0])
    T = np.array([[1, 0, tx * S], [0, 1, ty * S], [0, 0, 1]], dtype=np.float64)
    return T @ R @ C


def hsv_tables(gains: Sequence[float]) -> np.ndarray:
    """(3,256) u8 tables of RandomHSV: hue (x * r0) % 180, saturation / value clip(x * r, 0, 255)."""
    x = np.arange(256, dtype=np.float64)
    return np.stack([(x * gains[0]) % 180, np.clip(x * gains[1], 0, 255), np.clip(x * gains[2], 0, 255)]).astype(np.uint8)


def transform_boxes(boxes: np.ndarray, labels: np.ndarray, M: np.ndarray, scale: float, S: int, flip: bool):
    """xyxy boxes on the canvas -> boxes on the output: affine of the four corners, clip to [0,S], the candidate filter
    (both sides > 2 px, area kept > 10 %, aspect ratio < 100 - against the pre-affine box times `scale`), flip."""
    if len(boxes) == 0:
        return np.zeros((0, 4), np.float32), np.zeros((0,), np.int32)
    b = boxes.astype(np.float64)
    corners = np.stack([b[:, [0, 1]], b[:, [2, 3]], b[:, [0, 3]], b[:, [2, 1]]], axis=1)           # (n,4,2)
    pts = corners @ M[:2, :2].T + M[:2, 2]
    new = np.concatenate([pts.min(1), pts.max(1)], axis=1)
    new = np.clip(new, 0, S)
    w1, h1 = (b[:, 2] - b[:, 0]) * scale, (b[:, 3] - b[:, 1]) * scale
    w2, h2 = new[:, 2] - new[:, 0], new[:, 3] - new[:, 1]
    eps = 1e-16
    ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
    keep = (w2 > 2) & (h2 > 2) & (w2 * h2 / (w1 * h1 + eps) > 0.1) & (ar < 100)
    new, lab = new[keep], labels[keep]
    if flip:
        new = np.stack([S - new[:, 2], new[:, 1], S - new[:, 0], new[:, 3]], axis=1)
    return new.astype(np.float32), lab.astype(np.int32)


class DetAugment:
    """Draws one record per output image.  Seeded from Python's `random` unless a seed is given."""

    def __init__(self, size: int, seed: Optional[int] = None, mosaic: float = 1.0, hsv=(0.015, 0.7, 0.4),
                 fliplr: float = 0.5, translate: float = 0.1, scale: float = 0.5):
        self.S = int(size)
        self.rng = np.random.default_rng(random.getrandbits(63) if seed is None else seed)
        self.mosaic, self.hsv, self.fliplr, self.translate, self.scale = mosaic, hsv, fliplr, translate, scale

    def plan(self, index: int, n_dataset: int, use_mosaic: bool = True) -> dict:
        S, r = self.S, self.rng
        mosaic = bool(use_mosaic and r.random() < self.mosaic)
        p = dict(mosaic=mosaic, sources=[index])
        if mosaic:
            p["sources"] += [int(v) for v in r.integers(0, n_dataset, 3)]
            p["centre"] = (int(r.uniform(S / 2, 3 * S / 2)), int(r.uniform(S / 2, 3 * S / 2)))       # (xc, yc)
        p["scale"] = float(r.uniform(1 - self.scale, 1 + self.scale))
        p["translate"] = (float(r.uniform(0.5 - self.translate, 0.5 + self.translate)),
                          float(r.uniform(0.5 - self.translate, 0.5 + self.translate)))
        p["hsv"] = [float(v) for v in r.uniform(-1, 1, 3) * np.asarray(self.hsv) + 1]
        p["flip"] = bool(r.random() < self.fliplr)
        return p


def build_record(plan: dict, sizes: Sequence[Tuple[int, int]], tile_ids: Sequence[int], S: int):
    """plan + resized source sizes [(w,h)] + their tile slots -> rec_f (6) f32, rec_i (34) i32, lut (3,256) u8, the forward
    matrix and the per-source label offsets (padw, padh) on the canvas."""
    rec_i = np.zeros(34, dtype=np.int32)
    offs = []
    if plan["mosaic"]:
        canvas = 2 * S
        xc, yc = plan["centre"]
        for i, ((w, h), tid) in enumerate(zip(sizes, tile_ids)):
            (x1a, y1a, x2a, y2a), (x1b, y1b) = mosaic_placement(i, xc, yc, w, h, S)
            rec_i[2 + 8 * i:2 + 8 * i + 7] = (tid, x1a, y1a, x2a, y2a, x1b, y1b)
            offs.append((x1a - x1b, y1a - y1b))
        rec_i[0] = 4
    else:
        canvas = S
        (w, h), tid = sizes[0], tile_ids[0]
        left, top = int(round((S - w) / 2 - 0.1)), int(round((S - h) / 2 - 0.1))                         # LetterBox(center)
        rec_i[2:9] = (tid, left, top, left + w, top + h, 0, 0)
        offs.append((left, top))
        rec_i[0] = 1
    rec_i[1] = int(plan["flip"])
    M = affine_matrix(canvas, S, plan["scale"], *plan["translate"])
    rec_f = np.linalg.inv(M)[:2].reshape(-1).astype(np.float32)
    return rec_f, rec_i, hsv_tables(plan["hsv"]), M, offs, canvas


def augment_batch(samples: List[Tuple[str, str]], batch_idx: Sequence[int], aug: DetAugment, max_boxes: int, device: str,
                  use_mosaic: bool = True):
    """One training batch: decodes the planned sources, resizes them into tiles and composes the outputs on the device.
    -> images (B,S,S,3) u8 cuda, gt boxes (B,G,4) f32, gt labels (B,G) i32, counts (B) i32 (host tensors)."""
    import torch
    from PIL import Image
    from . import letterbox, mosaic_augment
    from .yolo_data import parse_label_text
    S, B = aug.S, len(batch_idx)
    plans = [aug.plan(int(i), len(samples), use_mosaic) for i in batch_idx]
    src_ids = sorted({s for p in plans for s in p["sources"]})
    slot = {s: k for k, s in enumerate(src_ids)}
    ims = [np.asarray(Image.open(samples[s][0]).convert("RGB")) for s in src_ids]
    Hc, Wc = max(im.shape[0] for im in ims), max(im.shape[1] for im in ims)
    canvas = np.zeros((len(ims), Hc, Wc, 3), dtype=np.uint8)
    geom = np.zeros((len(ims), 6), dtype=np.int32)
    sizes, labs = {}, {}
    for k, (s, im) in enumerate(zip(src_ids, ims)):
        h0, w0 = im.shape[:2]
        canvas[k, :h0, :w0] = im
        nw, nh = tile_geometry(w0, h0, S)
        geom[k] = (w0, h0, nw, nh, 0, 0)
        sizes[s] = (nw, nh)
        lp = samples[s][1]
        labs[s] = parse_label_text(open(lp).read()) if os.path.exists(lp) else np.zeros((0, 5))
    tiles = letterbox(torch.from_numpy(canvas).to(device), torch.from_numpy(geom).to(device), S)
    rec_f, rec_i, lut = np.zeros((B, 6), np.float32), np.zeros((B, 34), np.int32), np.zeros((B, 3, 256), np.uint8)
    boxes = np.zeros((B, max_boxes, 4), dtype=np.float32)
    labels = np.zeros((B, max_boxes), dtype=np.int32)
    counts = np.zeros((B,), dtype=np.int32)
    for b, p in enumerate(plans):
        srcs = p["sources"]
        rec_f[b], rec_i[b], lut[b], M, offs, cv = build_record(p, [sizes[s] for s in srcs], [slot[s] for s in srcs], S)
        bb, ll = [], []
        for s, (px, py) in zip(srcs, offs):
            lab, (w, h) = labs[s], sizes[s]
            if len(lab):
                xy = np.stack([(lab[:, 1] - lab[:, 3] / 2) * w + px, (lab[:, 2] - lab[:, 4] / 2) * h + py,
                               (lab[:, 1] + lab[:, 3] / 2) * w + px, (lab[:, 2] + lab[:, 4] / 2) * h + py], axis=1)
                bb.append(xy); ll.append(lab[:, 0])
        if bb:
            xy, lb = np.clip(np.concatenate(bb), 0, cv), np.concatenate(ll)
            good = (xy[:, 2] > xy[:, 0]) & (xy[:, 3] > xy[:, 1])                     # boxes clipped away by the canvas
            nb, nl = transform_boxes(xy[good], lb[good], M, p["scale"], S, p["flip"])
            n = min(len(nb), max_boxes)
            boxes[b, :n], labels[b, :n], counts[b] = nb[:n], nl[:n], n
    out = mosaic_augment(tiles, torch.from_numpy(rec_f).to(device), torch.from_numpy(rec_i).to(device),
                         torch.from_numpy(lut).to(device))
    return out, torch.from_numpy(boxes), torch.from_numpy(labels), torch.from_numpy(counts)
