"""Oracle (TEST INFRASTRUCTURE): box arithmetic of the hot path, on CPU.

Rows of SURVEY.md section 8(a) restated here: A5a, A5b, A6, B1, B2.
All fp arithmetic is IEEE f32, one rounding per operation (no FMA
contraction): torch CPU elementwise ops and numpy float32 both guarantee it.
"""
from __future__ import annotations

import math
import random as _random
from typing import List, Sequence, Tuple

import numpy as np
import torch


# --------------------------------------------------------------------------
# A5b  custom_nms  (README.md:62-84 == tech.md:72-94)                 pinned
# --------------------------------------------------------------------------
def box_iou(boxes1: torch.Tensor, boxes2: torch.Tensor) -> torch.Tensor:
    """Pairwise IoU, the form torchvision.ops.box_iou publishes (the import
    the README block relies on at README.md:77 is never shown; torchvision is
    absent from this image -> the formula is restated, f32, operation order
    area -> lt/rb -> clamp -> inter -> union -> divide).
    boxes: (n,4)/(m,4) xyxy f32 -> (n,m) f32.  0/0 -> NaN (kept as NaN)."""
    b1 = boxes1.to(torch.float32)
    b2 = boxes2.to(torch.float32)
    area1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    area2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[None, :, :2])
    rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = (area1[:, None] + area2[None, :]) - inter
    return inter / union


def score_order(scores: torch.Tensor) -> torch.Tensor:
    """Descending-score order with the tie rule this build DEFINES:
    equal scores keep ascending original index (README.md:64 uses
    torch.argsort(descending=True), whose tie order is unspecified)."""
    return torch.sort(scores.to(torch.float32), descending=True, stable=True).indices


def custom_nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float = 0.45) -> List[int]:
    """Restates README.md:62-84 statement by statement: class-agnostic greedy
    NMS, no score threshold, no max-det, a candidate survives a kept box iff
    IoU < thr (strict; NaN does not survive), returns ORIGINAL indices in
    descending-score order.  n=0 -> [], n=1 -> [0]."""
    sorted_indices = score_order(scores)
    thr = torch.tensor(iou_threshold, dtype=torch.float32)
    keep: List[int] = []
    while sorted_indices.numel() > 0:
        i = int(sorted_indices[0])
        keep.append(i)
        if sorted_indices.numel() == 1:
            break
        ious = box_iou(boxes[i:i + 1], boxes[sorted_indices[1:]])
        mask = ious.reshape(-1) < thr
        sorted_indices = sorted_indices[1:][mask]
    return keep


# --------------------------------------------------------------------------
# A5a  EfficientNMS_TRT layout (tech.md:41-47, test.ipynb:20-24)    unpinned
# --------------------------------------------------------------------------
def efficient_nms(boxes: torch.Tensor, scores: torch.Tensor, score_threshold: float = 0.25,
                  iou_threshold: float = 0.65, max_output_boxes: int = 100,
                  pre_nms_topk: int = 4096):
    """Per-class greedy NMS producing the engine's 4 outputs (KAT-2,
    test.ipynb:20-24): num_dets (B,1) i32, bboxes (B,K,4) f32, scores (B,K)
    f32, labels (B,K) i32, zero padded, sorted by score.

    Parity UNPINNED: the TensorRT plugin is absent; semantics follow the
    parameters at tech.md:41-47 / docs/YOLO_TensorRT_Technical.md:109-150 and
    the prose at :137-146 (suppress when IoU > thr).  Defined here:
    candidate = (anchor a, class c) with score > score_threshold (strict, as
    docs/YOLO_TensorRT_Technical.md:252); candidates ordered by (score desc,
    flat index a*nc+c asc); only the first ``pre_nms_topk`` enter the greedy
    scan (the plugin's bounded selection); a candidate is suppressed by an
    already kept one of the SAME class when IoU > iou_threshold."""
    B, A, nc = scores.shape
    K = max_output_boxes
    num = torch.zeros(B, 1, dtype=torch.int32)
    ob = torch.zeros(B, K, 4, dtype=torch.float32)
    osc = torch.zeros(B, K, dtype=torch.float32)
    ol = torch.zeros(B, K, dtype=torch.int32)
    thr = torch.tensor(iou_threshold, dtype=torch.float32)
    for b in range(B):
        flat = scores[b].reshape(-1).to(torch.float32)
        cand = torch.nonzero(flat > score_threshold).reshape(-1)
        if cand.numel() == 0:
            continue
        order = torch.sort(flat[cand], descending=True, stable=True).indices
        cand = cand[order][:pre_nms_topk]
        cb = boxes[b, cand // nc].to(torch.float32)
        cs = flat[cand]
        cl = (cand % nc).to(torch.int32)
        alive = torch.ones(cand.numel(), dtype=torch.bool)
        kept = 0
        for i in range(cand.numel()):
            if not alive[i]:
                continue
            ob[b, kept] = cb[i]
            osc[b, kept] = cs[i]
            ol[b, kept] = cl[i]
            kept += 1
            if kept == K:
                break
            if i + 1 < cand.numel():
                iou = box_iou(cb[i:i + 1], cb[i + 1:]).reshape(-1)
                sup = (iou > thr) & (cl[i + 1:] == cl[i])
                alive[i + 1:] &= ~sup
        num[b, 0] = kept
    return num, ob, osc, ol


# --------------------------------------------------------------------------
# A6  det_postprocess + coordinate restore                          unpinned
#     (YOLOTensorRT_yolodet_py_解读.md:82-99)
# --------------------------------------------------------------------------
def restore_and_filter(num_dets: int, bboxes: torch.Tensor, scores: torch.Tensor, labels: torch.Tensor,
                       ratio: float, dwdh: Tuple[float, float], conf: float = 0.35,
                       coord_mode: str = "trunc"):
    """slice [:num_dets]; bboxes -= (dw,dh,dw,dh); bboxes /= ratio (f32, in
    that order); drop score < conf; coordinates -> int.  The walkthrough only
    says "converted to an integer list" (解读.md:96): ``trunc`` = Python int()
    (toward zero), ``round`` = torch.round (half to even) then int.
    Returns (idx list into the 100 slots, int boxes (m,4), scores, labels)."""
    n = int(num_dets)
    bb = bboxes[:n].to(torch.float32).clone()
    d = torch.tensor([dwdh[0], dwdh[1], dwdh[0], dwdh[1]], dtype=torch.float32)
    bb = (bb - d) / torch.tensor(ratio, dtype=torch.float32)
    idx, ib, sc, lb = [], [], [], []
    for j in range(n):
        if float(scores[j]) < conf:
            continue
        v = bb[j].round() if coord_mode == "round" else bb[j]
        ib.append([int(x) for x in v.tolist()])
        idx.append(j)
        sc.append(float(scores[j]))
        lb.append(int(labels[j]))
    return idx, ib, sc, lb


# --------------------------------------------------------------------------
# B1  crop_image integer inflate  (utils/trainClass.py:70-93)         pinned
# --------------------------------------------------------------------------
def inflate_eval(x_min: int, y_min: int, x_max: int, y_max: int, width: int, height: int):
    """utils/trainClass.py:76-77,85-91: Python floor division, eval branch."""
    dis_x = (x_max - x_min) // 10
    dis_y = (y_max - y_min) // 10
    x_max = min(width, x_max + dis_x // 2)
    x_min = max(0, x_min - dis_x // 2)
    y_max = min(height, y_max + dis_y // 2)
    y_min = max(0, y_min - dis_y // 2)
    return x_min, y_min, x_max, y_max


def inflate_train(x_min: int, y_min: int, x_max: int, y_max: int, width: int, height: int,
                  rng: _random.Random):
    """utils/trainClass.py:78-84: draw order x_max, x_min, y_max, y_min."""
    dis_x = (x_max - x_min) // 10
    dis_y = (y_max - y_min) // 10
    x_max = min(width, x_max + rng.randint(0, dis_x))
    x_min = max(0, x_min - rng.randint(0, dis_x))
    y_max = min(height, y_max + rng.randint(0, dis_y))
    y_min = max(0, y_min - rng.randint(0, dis_y))
    return x_min, y_min, x_max, y_max


# --------------------------------------------------------------------------
# B2  eval transform  (utils/trainClass.py:218-221,265-266; app.py:39-42)
#     cv2 / albumentations absent -> index rule and rounding       unpinned
# --------------------------------------------------------------------------
def nearest_index_table(dst: int, src: int) -> np.ndarray:
    """OpenCV INTER_NEAREST source index for each destination index:
    fx = dst/src (double); ifx = 1/fx; s = min(floor(d*ifx), src-1)."""
    fx = float(dst) / float(src)
    ifx = 1.0 / fx
    return np.array([min(int(math.floor(d * ifx)), src - 1) for d in range(dst)], dtype=np.int32)


NORM_MEAN = np.float32(0.5) * np.float32(255.0)
NORM_RCP = np.reciprocal(np.float32(0.5) * np.float32(255.0), dtype=np.float32)


def normalize_u8(x_u8: np.ndarray) -> np.ndarray:
    """albumentations Normalize(mean=.5,std=.5,max_pixel_value=255) in f32:
    (x - 127.5) * fl32(1/127.5), two roundings."""
    x = x_u8.astype(np.float32)
    x = x - NORM_MEAN
    return x * NORM_RCP


def crop_resize_normalize(img_hwc_u8: np.ndarray, box: Sequence[int], out_hw=(224, 224)) -> np.ndarray:
    """PIL crop (right/bottom exclusive, utils/trainClass.py:92) -> nearest
    resize to out_hw -> normalize -> CHW f32 (utils/trainClass.py:265-266)."""
    x0, y0, x1, y1 = [int(v) for v in box]
    h, w = y1 - y0, x1 - x0
    assert h > 0 and w > 0, "degenerate crop"
    ty = nearest_index_table(out_hw[0], h) + y0
    tx = nearest_index_table(out_hw[1], w) + x0
    g = img_hwc_u8[ty][:, tx]                      # (224,224,3)
    return np.ascontiguousarray(np.transpose(normalize_u8(g), (2, 0, 1)))


def patchify(chw: np.ndarray, patch: int) -> np.ndarray:
    """(3,H,W) -> (H/P * W/P, 3*P*P) rows ordered like a conv k=s=P weight
    flattened (c, py, px): the A operand of the patch-embed GEMM."""
    c, h, w = chw.shape
    gh, gw = h // patch, w // patch
    x = chw.reshape(c, gh, patch, gw, patch)
    return np.ascontiguousarray(x.transpose(1, 3, 0, 2, 4).reshape(gh * gw, c * patch * patch))


# --------------------------------------------------------------------------
# A1  letterbox geometry (YOLOTensorRT_yolodet_py_解读.md:67-69)     unpinned
# --------------------------------------------------------------------------
def letterbox_params(h: int, w: int, new_w: int = 640, new_h: int = 640):
    """Published upstream form: r = min(new_h/h, new_w/w); unpad = round(w*r),
    round(h*r); (dw,dh) = half the remaining pad; borders round(d -/+ 0.1)."""
    r = min(new_h / h, new_w / w)
    nw, nh = int(round(w * r)), int(round(h * r))
    dw, dh = (new_w - nw) / 2.0, (new_h - nh) / 2.0
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return r, (dw, dh), (nw, nh), (top, bottom, left, right)


def letterbox_image(img_hwc_u8: np.ndarray, new_w: int = 640, new_h: int = 640, fill: int = 114) -> np.ndarray:
    """A1 resample (YOLOTensorRT_yolodet_py_解读.md:67-69: `letterbox(bgr, (W, H))` = aspect-preserving resize +
    constant border).  Parity UNPINNED: the reference resizes with cv2.resize(INTER_LINEAR), whose fixed-point
    arithmetic is not reproducible without OpenCV (absent).  This build's stated rule, restated here in numpy float32
    with one rounding per operation (the device kernel is compiled with -ffp-contract=off, so the two agree bit for bit):
        src = (dst + 0.5) * (src_size / dst_size) - 0.5, clamped to [0, src_size - 1]  (half-pixel centres)
        v   = (a + (b - a) fx) (1 - fy) + (c + (d - c) fx) fy,  out = floor(v + 0.5)   (round half up)
    identity when the resized size equals the source size; border value 114."""
    h, w = img_hwc_u8.shape[:2]
    _, _, (nw, nh), (top, _, left, _) = letterbox_params(h, w, new_w, new_h)
    out = np.full((new_h, new_w, 3), fill, dtype=np.uint8)
    if (nw, nh) == (w, h):
        out[top:top + nh, left:left + nw] = img_hwc_u8
        return out
    f32 = np.float32
    half = f32(0.5)

    def taps(n_dst, n_src):
        s = (np.arange(n_dst, dtype=np.float32) + half) * (f32(n_src) / f32(n_dst)) - half
        s = np.minimum(np.maximum(s, f32(0)), f32(n_src - 1))
        i0 = s.astype(np.int32)
        i1 = np.minimum(i0 + 1, n_src - 1)
        return i0, i1, (s - i0.astype(np.float32)).astype(np.float32)

    x0, x1, fx = taps(nw, w)
    y0, y1, fy = taps(nh, h)
    im = img_hwc_u8.astype(np.float32)
    fx_ = fx[None, :, None]
    fy_ = fy[:, None, None]
    a, b = im[y0][:, x0], im[y0][:, x1]
    c, d = im[y1][:, x0], im[y1][:, x1]
    v = (a + (b - a) * fx_) * (f32(1) - fy_) + (c + (d - c) * fx_) * fy_
    res = np.clip(np.floor(v + half), 0, 255).astype(np.uint8)
    out[top:top + nh, left:left + nw] = res
    return out
