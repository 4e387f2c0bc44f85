"""TEST INFRASTRUCTURE ONLY (imported by tests/): CPU restatement of `yv_mosaic_augment` (csrc/augment.hip) for one
record, plus an explicit-canvas statement of Mosaic for the integer cases.

Path: the default augmentation of `model.train()` (utils/trainYolo.py:28): Mosaic -> RandomPerspective -> RandomHSV ->
RandomFlip.  All of it lives in `ultralytics` / OpenCV, absent from the reference tree and this image: PARITY UNPINNED
against the library; the restatement follows its published algorithm with this build's own 8-bit HSV rounding.
"""
import numpy as np

FILL = 114.0


def _canvas_lookup(tiles, rec_i, S, i, j, c):
    """Value of canvas pixel (i, j), channel c (arrays of any equal shape)."""
    out = np.full(i.shape, FILL, dtype=np.float32)
    done = np.zeros(i.shape, dtype=bool)
    nt = int(np.clip(rec_i[0], 0, 4))
    for t in range(nt):
        tid, x1a, y1a, x2a, y2a, x1b, y1b = (int(v) for v in rec_i[2 + 8 * t:2 + 8 * t + 7])
        inside = (i >= x1a) & (i < x2a) & (j >= y1a) & (j < y2a) & ~done
        sx, sy = i - x1a + x1b, j - y1a + y1b
        ok = inside & (sx >= 0) & (sx < S) & (sy >= 0) & (sy < S) & (0 <= tid < tiles.shape[0])
        out[ok] = tiles[min(max(tid, 0), tiles.shape[0] - 1)][sy[ok], sx[ok], c].astype(np.float32)
        done |= inside
    return out


def _rhu(v):
    return np.floor((v + np.float32(0.5)).astype(np.float32))


def apply_record(tiles, rec_f, rec_i, lut, S):
    """tiles (N,S,S,3) u8 -> (S,S,3) u8; f32 arithmetic, one rounding per operation, order of csrc/augment.hip."""
    f = np.float32
    a = np.asarray(rec_f, dtype=f)
    y, x = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
    xs = (S - 1 - x if int(rec_i[1]) else x).astype(f)
    ys = y.astype(f)
    with np.errstate(invalid="ignore", over="ignore"):
        u = ((a[0] * xs + a[1] * ys) + a[2]).astype(f)
        v = ((a[3] * xs + a[4] * ys) + a[5]).astype(f)
    lim = f(8 * S)
    u = np.where(np.isnan(u), -lim, np.clip(u, -lim, lim)).astype(f)
    v = np.where(np.isnan(v), -lim, np.clip(v, -lim, lim)).astype(f)
    uf, vf = np.floor(u), np.floor(v)
    fx, fy = (u - uf).astype(f), (v - vf).astype(f)
    gx, gy = (f(1) - fx).astype(f), (f(1) - fy).astype(f)
    i0, j0 = uf.astype(np.int64), vf.astype(np.int64)
    rgb = []
    for c in range(3):
        v00, v01 = _canvas_lookup(tiles, rec_i, S, i0, j0, c), _canvas_lookup(tiles, rec_i, S, i0 + 1, j0, c)
        v10, v11 = _canvas_lookup(tiles, rec_i, S, i0, j0 + 1, c), _canvas_lookup(tiles, rec_i, S, i0 + 1, j0 + 1, c)
        top = (v00 * gx + v01 * fx).astype(f)
        bot = (v10 * gx + v11 * fx).astype(f)
        rgb.append(np.clip(_rhu((top * gy + bot * fy).astype(f)), 0, 255).astype(f))
    R, G, B = rgb
    vmax, vmin = np.maximum(R, np.maximum(G, B)), np.minimum(R, np.minimum(G, B))
    diff = (vmax - vmin).astype(f)
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.where(vmax > 0, _rhu(((f(255) * diff).astype(f) / vmax).astype(f)), f(0)).astype(f)
        hr = ((f(60) * (G - B).astype(f)).astype(f) / diff).astype(f)
        hg = (f(120) + ((f(60) * (B - R).astype(f)).astype(f) / diff).astype(f)).astype(f)
        hb = (f(240) + ((f(60) * (R - G).astype(f)).astype(f) / diff).astype(f)).astype(f)
    h = np.where(vmax == R, hr, np.where(vmax == G, hg, hb))
    h = np.where(diff > 0, h, f(0)).astype(f)
    h = np.where(h < 0, (h + f(360)).astype(f), h).astype(f)
    h8 = _rhu((h * f(0.5)).astype(f)).astype(np.int64)
    h8 = np.where(h8 >= 180, h8 - 180, h8)
    lut = np.asarray(lut)
    H2, S2, V2 = lut[0][h8].astype(f), lut[1][s.astype(np.int64)].astype(f), lut[2][vmax.astype(np.int64)].astype(f)
    hs = (H2 / f(30)).astype(f)
    sec = np.floor(hs)
    fr = (hs - sec).astype(f)
    sn = (S2 / f(255)).astype(f)
    pp = (V2 * (f(1) - sn).astype(f)).astype(f)
    qq = (V2 * (f(1) - (sn * fr).astype(f)).astype(f)).astype(f)
    tt = (V2 * (f(1) - (sn * (f(1) - fr).astype(f)).astype(f)).astype(f)).astype(f)
    si = sec.astype(np.int64) % 6
    r2 = np.choose(si, [V2, qq, pp, pp, tt, V2])
    g2 = np.choose(si, [tt, V2, V2, qq, pp, pp])
    b2 = np.choose(si, [pp, pp, tt, V2, V2, qq])
    out = np.stack([np.clip(_rhu(t), 0, 255) for t in (r2, g2, b2)], axis=-1)
    return out.astype(np.uint8)


def explicit_mosaic(resized, centre, S):
    """Builds the 2S x 2S canvas the way Mosaic is published: four resized images pasted around `centre`, 114 elsewhere.
    resized: four (h,w,3) u8 arrays.  Returns the canvas and the per-image (padw, padh)."""
    xc, yc = centre
    s2 = 2 * S
    canvas = np.full((s2, s2, 3), 114, dtype=np.uint8)
    pads = []
    for i, im in enumerate(resized):
        h, w = im.shape[:2]
        if i == 0:
            x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
            x1b, y1b, x2b, y2b = w - (x2a - x1a), h - (y2a - y1a), w, h
        elif i == 1:
            x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, s2), yc
            x1b, y1b, x2b, y2b = 0, h - (y2a - y1a), min(w, x2a - x1a), h
        elif i == 2:
            x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(s2, yc + h)
            x1b, y1b, x2b, y2b = w - (x2a - x1a), 0, w, min(y2a - y1a, h)
        else:
            x1a, y1a, x2a, y2a = xc, yc, min(xc + w, s2), min(s2, yc + h)
            x1b, y1b, x2b, y2b = 0, 0, min(w, x2a - x1a), min(y2a - y1a, h)
        canvas[y1a:y2a, x1a:x2a] = im[y1b:y2b, x1b:x2b]
        pads.append((x1a - x1b, y1a - y1b))
    return canvas, pads
