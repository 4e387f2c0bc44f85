"""TEST INFRASTRUCTURE ONLY (imported by tests/): CPU restatement of what `yv_augment_patchify` does with ONE
augmentation record, plus a stage-by-stage statement of the reference's training transform for the cases that
need no interpolation.

Path: data_transforms['train'] after Resize + Normalize, utils/trainClass.py:201-216 (HorizontalFlip, RandomCrop +
PadIfNeeded, ShiftScaleRotate, ChannelShuffle, GridDistortion | ElasticTransform, CoarseDropout).  The transforms
live in albumentations / OpenCV, which are not in this image and not vendored by the reference: PARITY UNPINNED
against the library; the restatement follows its published 1.x behaviour (see yvhip/augment.py for the conventions).
"""
import numpy as np
import torch


def _reflect101(i, n):
    if n == 1:
        return np.zeros_like(i)
    per = 2 * n - 2
    m = np.mod(i, per)
    return np.where(m < n, m, per - m)


def apply_record(x, geo, idx, P):
    """x (3,S,S) f32, geo (6+2S) f32, idx (36+2S) i32 -> (g*g, 3*P*P) f32 holding bf16-rounded values.
    Every operation is one f32 rounding, in the order csrc/augment.hip uses."""
    f = np.float32
    x = np.asarray(x, dtype=f)
    S = x.shape[1]
    a = np.asarray(geo[:6], dtype=f)
    lutx, luty = np.asarray(geo[6:6 + S], dtype=f), np.asarray(geo[6 + S:6 + 2 * S], dtype=f)
    perm = np.clip(idx[0:3], 0, 2)
    nh = int(np.clip(idx[3], 0, 8))
    holes = np.asarray(idx[4:36]).reshape(8, 4)
    mapx = np.clip(idx[36:36 + S], 0, S - 1)
    mapy = np.clip(idx[36 + S:36 + 2 * S], 0, S - 1)
    cx, cy = lutx[None, :], luty[:, None]
    with np.errstate(invalid="ignore", over="ignore"):              # hostile records: NaN / inf are clamped below
        u = (a[0] * cx + a[1] * cy) + a[2]
        v = (a[3] * cx + a[4] * cy) + a[5]
    lim = f(4 * S)
    u = np.where(np.isnan(u), -lim, np.clip(u, -lim, lim)).astype(f)
    v = np.where(np.isnan(v), -lim, np.clip(v, -lim, lim)).astype(f)
    uf, vf = np.floor(u), np.floor(v)
    fx, fy = (u - uf).astype(f), (v - vf).astype(f)
    ix, iy = uf.astype(np.int64), vf.astype(np.int64)
    x0, x1 = mapx[_reflect101(ix, S)], mapx[_reflect101(ix + 1, S)]
    y0, y1 = mapy[_reflect101(iy, S)], mapy[_reflect101(iy + 1, S)]
    gx1, gy1 = (f(1) - fx).astype(f), (f(1) - fy).astype(f)
    out = np.empty((3, S, S), dtype=f)
    for c in range(3):
        pl = x[perm[c]]
        top = (pl[y0, x0] * gx1 + pl[y0, x1] * fx).astype(f)
        bot = (pl[y1, x0] * gx1 + pl[y1, x1] * fx).astype(f)
        out[c] = (top * gy1 + bot * fy).astype(f)
    for h in range(nh):
        qx1, qy1, qx2, qy2 = (int(t) for t in holes[h])
        out[:, max(qy1, 0):max(qy2, 0), max(qx1, 0):max(qx2, 0)] = 0
    g = S // P
    rows = out.reshape(3, g, P, g, P).transpose(1, 3, 0, 2, 4).reshape(g * g, 3 * P * P)
    return torch.from_numpy(np.ascontiguousarray(rows)).to(torch.bfloat16).to(torch.float32).numpy()


def sequential_integer(x, flip=False, crop_xy=None, crop=200, shift=(0, 0), perm=(0, 1, 2), holes=()):
    """Stage-by-stage statement with array operations only (no tables, no inverse maps), for parameter sets that need
    no interpolation: flip -> crop + centred reflect padding -> integer translation with reflected border -> channel
    shuffle -> holes.  x (3,S,S) -> (3,S,S)."""
    x = np.asarray(x, dtype=np.float32)
    S = x.shape[1]
    if flip:
        x = x[:, :, ::-1]
    if crop_xy is not None:
        x1, y1 = crop_xy
        x = x[:, y1:y1 + crop, x1:x1 + crop]
        lo = int((S - crop) / 2.0)
        hi = S - crop - lo
        x = np.pad(x, ((0, 0), (lo, hi), (lo, hi)), mode="reflect")
    sx, sy = shift                                                  # dst(x, y) = src(x - sx, y - sy)
    if sx or sy:
        m = max(abs(sx), abs(sy))
        xp = np.pad(x, ((0, 0), (m, m), (m, m)), mode="reflect")
        x = xp[:, m - sy:m - sy + S, m - sx:m - sx + S]
    x = x[list(perm)]
    x = x.copy()
    for qx1, qy1, qx2, qy2 in holes:
        x[:, qy1:qy2, qx1:qx2] = 0
    return x
