"""Host-side YOLO-format dataset reader for `utils.trainYolo.train` (the `data=` yaml of utils/trainYolo.py:33 and the
`images/{train,val}` + `labels/{train,val}` tree that utils/class_config.py:89-148 writes).

Deliberately minimal (SURVEY.md 8(f) N4 - mosaic / HSV / flips of the ultralytics loader are not built): images are
letterboxed to the square network input with the published letterbox arithmetic and labels are mapped into those
pixels.  Label files may carry the reference writer's literal backslash-n separators (utils/class_config.py:84)."""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np
import torch

IMG_EXT = (".jpg", ".jpeg", ".png", ".bmp")


def read_data_yaml(path: str) -> Dict:
    import yaml
    with open(path) as f:
        cfg = yaml.safe_load(f)
    if not isinstance(cfg, dict) or "train" not in cfg:
        raise ValueError(f"{path}: expected a mapping with a 'train' entry")
    root = cfg.get("path") or os.path.dirname(os.path.abspath(path))
    names = cfg.get("names")
    if isinstance(names, dict):
        names = [names[k] for k in sorted(names)]
    nc = int(cfg.get("nc", len(names) if names else 0))
    res = lambda p: p if os.path.isabs(p) else os.path.join(root, p)
    return {"train": res(cfg["train"]), "val": res(cfg["val"]) if cfg.get("val") else None, "nc": nc, "names": names}


def list_samples(image_dir: str) -> List[Tuple[str, str]]:
    """[(image path, label path)] sorted by name; labels live in the sibling `labels/<split>` directory."""
    out = []
    for name in sorted(os.listdir(image_dir)):
        if not name.lower().endswith(IMG_EXT):
            continue
        img = os.path.join(image_dir, name)
        sep = os.sep + "images" + os.sep
        i = img.rfind(sep)
        lab_dir = (img[:i] + os.sep + "labels" + os.sep + os.path.dirname(img[i + len(sep):])) if i >= 0 else image_dir
        out.append((img, os.path.join(lab_dir, os.path.splitext(name)[0] + ".txt")))
    return out


def parse_label_text(text: str) -> np.ndarray:
    """'cls cx cy w h' records -> (n, 5) float array; records are separated by newlines or by the literal two
    characters backslash + n that the reference's writeTxt emits."""
    rows = []
    for rec in text.replace("\\n", "\n").splitlines():
        parts = rec.split()
        if len(parts) >= 5:
            rows.append([float(v) for v in parts[:5]])
    return np.asarray(rows, dtype=np.float64).reshape(-1, 5)


def letterbox_host(im: np.ndarray, size: int, color: int = 114):
    """(h,w,3) u8 -> (size,size,3) u8, ratio, (left, top): r = min(S/h, S/w); unpad = round(w*r), round(h*r);
    pad = round(d - 0.1) (the arithmetic of YOLOTensorRT_yolodet_py_解读.md:67-69); bilinear resize (PIL)."""
    from PIL import Image
    h, w = im.shape[:2]
    r = min(size / h, size / w)
    nw, nh = int(round(w * r)), int(round(h * r))
    dw, dh = (size - nw) / 2, (size - nh) / 2
    top, left = int(round(dh - 0.1)), int(round(dw - 0.1))
    if (nw, nh) != (w, h):
        im = np.asarray(Image.fromarray(im).resize((nw, nh), Image.BILINEAR))
    out = np.full((size, size, 3), color, dtype=np.uint8)
    out[top:top + nh, left:left + nw] = im
    return out, r, (left, top)


def load_batch(samples: List[Tuple[str, str]], size: int, max_boxes: int):
    """-> images (B,size,size,3) u8, gt_boxes (B,G,4) f32 xyxy letterboxed pixels, gt_labels (B,G) i32, counts (B) i32."""
    from PIL import Image
    B = len(samples)
    imgs = np.zeros((B, size, size, 3), dtype=np.uint8)
    boxes = np.zeros((B, max_boxes, 4), dtype=np.float32)
    labels = np.zeros((B, max_boxes), dtype=np.int32)
    counts = np.zeros((B,), dtype=np.int32)
    for i, (ip, lp) in enumerate(samples):
        im = np.asarray(Image.open(ip).convert("RGB"))
        h, w = im.shape[:2]
        imgs[i], r, (left, top) = letterbox_host(im, size)
        lab = parse_label_text(open(lp).read()) if os.path.exists(lp) else np.zeros((0, 5))
        n = min(len(lab), max_boxes)
        for j in range(n):
            c, cx, cy, bw, bh = lab[j]
            x1, y1, x2, y2 = (cx - bw / 2) * w, (cy - bh / 2) * h, (cx + bw / 2) * w, (cy + bh / 2) * h
            boxes[i, j] = [x1 * r + left, y1 * r + top, x2 * r + left, y2 * r + top]
            labels[i, j] = int(c)
        counts[i] = n
    return torch.from_numpy(imgs), torch.from_numpy(boxes), torch.from_numpy(labels), torch.from_numpy(counts)


def max_boxes_per_image(samples: List[Tuple[str, str]]) -> int:
    m = 1
    for _, lp in samples:
        if os.path.exists(lp):
            m = max(m, len(parse_label_text(open(lp).read())))
    return m
