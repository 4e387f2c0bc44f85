"""Oracle (TEST INFRASTRUCTURE): fp32 CPU forward of the YOLOv8 detector, rows
A2-A4 of SURVEY.md section 8(a).

Parity UNPINNED: the arithmetic lives in `ultralytics` (not listed in
requirements.txt; call sites utils/trainYolo.py:1,13,21,33) which is absent
from the reference tree and from this image.  The topology is anchored on the
reference's own artefacts: KAT-1 "168 layers, 3006623 parameters, 8.1 GFLOPs"
(test.ipynb:12), the TensorRT layer names (test.ipynb:25-1285), the TRT
builders quoted at docs/YOLO_TensorRT_Technical.md:160-212 (Conv = conv +
folded-BN bias + SiLU, C2f split/concat) and the decode at :14-30,72-77."""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768)}
REG_MAX = 16


def _ch(c: int, width: float, max_ch: int) -> int:
    return int(math.ceil(min(c, max_ch) * width / 8) * 8)


def _rep(n: int, depth: float) -> int:
    return max(round(n * depth), 1)


def topology(scale: str = "n"):
    """[(idx, kind, args)] for the 23 modules of yolov8.yaml at this scale."""
    d, w, m = SCALES[scale]
    c = lambda x: _ch(x, w, m)
    r = lambda x: _rep(x, d)
    return [
        (0, "conv", (3, c(64), 3, 2)), (1, "conv", (c(64), c(128), 3, 2)),
        (2, "c2f", (c(128), c(128), r(3), True)), (3, "conv", (c(128), c(256), 3, 2)),
        (4, "c2f", (c(256), c(256), r(6), True)), (5, "conv", (c(256), c(512), 3, 2)),
        (6, "c2f", (c(512), c(512), r(6), True)), (7, "conv", (c(512), c(1024), 3, 2)),
        (8, "c2f", (c(1024), c(1024), r(3), True)), (9, "sppf", (c(1024), c(1024))),
        (10, "up", ()), (11, "cat", (6,)), (12, "c2f", (c(1024) + c(512), c(512), r(3), False)),
        (13, "up", ()), (14, "cat", (4,)), (15, "c2f", (c(512) + c(256), c(256), r(3), False)),
        (16, "conv", (c(256), c(256), 3, 2)), (17, "cat", (12,)),
        (18, "c2f", (c(256) + c(512), c(512), r(3), False)),
        (19, "conv", (c(512), c(512), 3, 2)), (20, "cat", (9,)),
        (21, "c2f", (c(512) + c(1024), c(1024), r(3), False)),
        (22, "detect", ((c(256), c(512), c(1024)),)),
    ]


def conv_shapes(scale: str = "n", nc: int = 5) -> List[Tuple[str, int, int, int, int]]:
    """Every conv of the fused model: (key prefix, cin, cout, k, stride)."""
    out = []
    for idx, kind, a in topology(scale):
        p = f"model.{idx}."
        if kind == "conv":
            out.append((p + "conv", a[0], a[1], a[2], a[3]))
        elif kind == "c2f":
            c1, c2, n, _ = a
            c = c2 // 2
            out.append((p + "cv1.conv", c1, 2 * c, 1, 1))
            out.append((p + "cv2.conv", (2 + n) * c, c2, 1, 1))
            for j in range(n):
                out.append((p + f"m.{j}.cv1.conv", c, c, 3, 1))
                out.append((p + f"m.{j}.cv2.conv", c, c, 3, 1))
        elif kind == "sppf":
            c1, c2 = a
            out.append((p + "cv1.conv", c1, c1 // 2, 1, 1))
            out.append((p + "cv2.conv", c1 * 2, c2, 1, 1))
        elif kind == "detect":
            ch = a[0]
            c2 = max(16, ch[0] // 4, REG_MAX * 4)
            c3 = max(ch[0], min(nc, 100))
            for s, ci in enumerate(ch):
                out.append((p + f"cv2.{s}.0.conv", ci, c2, 3, 1))
                out.append((p + f"cv2.{s}.1.conv", c2, c2, 3, 1))
                out.append((p + f"cv2.{s}.2", c2, 4 * REG_MAX, 1, 1))
                out.append((p + f"cv3.{s}.0.conv", ci, c3, 3, 1))
                out.append((p + f"cv3.{s}.1.conv", c3, c3, 3, 1))
                out.append((p + f"cv3.{s}.2", c3, nc, 1, 1))
    return out


def param_count(scale: str = "n", nc: int = 5) -> int:
    """Fused (conv+bias) parameters + the 16 DFL constants: KAT-1."""
    return sum(ci * co * k * k + co for _, ci, co, k, _ in conv_shapes(scale, nc)) + REG_MAX


def macs(scale: str = "n", nc: int = 5, size: int = 640) -> int:
    """Multiply-accumulates of one forward at size x size (KAT-1: 2*MAC = 8.1 GFLOP)."""
    res: Dict[int, int] = {}
    total = 0
    h = size
    for idx, kind, a in topology(scale):
        if kind == "conv":
            h = h // a[3]
            total += a[0] * a[1] * a[2] * a[2] * h * h
        elif kind == "c2f":
            c1, c2, n, _ = a
            c = c2 // 2
            total += (c1 * 2 * c + (2 + n) * c * c2 + n * 2 * 9 * c * c) * h * h
        elif kind == "sppf":
            total += (a[0] * (a[0] // 2) + 2 * a[0] * a[1]) * h * h
        elif kind == "up":
            h *= 2
        elif kind == "cat":
            pass
        elif kind == "detect":
            ch = a[0]
            c2 = max(16, ch[0] // 4, REG_MAX * 4)
            c3 = max(ch[0], min(nc, 100))
            for s, ci in enumerate(ch):
                hs = size // (8 << s)
                total += (ci * c2 * 9 + c2 * c2 * 9 + c2 * 64 + ci * c3 * 9 + c3 * c3 * 9 + c3 * nc) * hs * hs
        res[idx] = h
    return total


def init_state(scale: str = "n", nc: int = 5, seed: int = 42, cls_bias: float = 0.0) -> Dict[str, torch.Tensor]:
    """Seeded Kaiming-style fused weights (BN folded to identity), with the
    ultralytics key layout of a fused model.  Gains are chosen so activations
    stay O(1) through SiLU stacks."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for key, ci, co, k, _ in conv_shapes(scale, nc):
        fan = ci * k * k
        sd[key + ".weight"] = torch.randn(co, ci, k, k, generator=g) * math.sqrt(2.0 / fan)
        sd[key + ".bias"] = torch.randn(co, generator=g) * 0.1
        if key.startswith("model.22.cv3.") and key.endswith(".2"):
            sd[key + ".bias"] = sd[key + ".bias"] + cls_bias
    sd["model.22.dfl.conv.weight"] = torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1)
    return sd


def _conv(sd, key, x, k, s, act=True):
    y = F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], stride=s, padding=k // 2)
    return F.silu(y) if act else y


def _c2f(sd, p, x, n, shortcut):
    y = list(_conv(sd, p + "cv1.conv", x, 1, 1).chunk(2, 1))
    for j in range(n):
        t = _conv(sd, p + f"m.{j}.cv2.conv", _conv(sd, p + f"m.{j}.cv1.conv", y[-1], 3, 1), 3, 1)
        y.append(y[-1] + t if shortcut else t)
    return _conv(sd, p + "cv2.conv", torch.cat(y, 1), 1, 1)


def make_anchors(size: int = 640):
    """docs/YOLO_TensorRT_Technical.md:14-30: cell centres (+0.5), order y*w+x,
    scales 8,16,32 concatenated.  Returns anchors (A,2) and strides (A,)."""
    pts, st = [], []
    for s in (8, 16, 32):
        w = h = size // s
        sx = torch.arange(w, dtype=torch.float32) + 0.5
        sy = torch.arange(h, dtype=torch.float32) + 0.5
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((xx, yy), -1).view(-1, 2))
        st.append(torch.full((h * w,), float(s)))
    return torch.cat(pts), torch.cat(st)


def decode(raw: torch.Tensor, nc: int, size: int = 640):
    """raw (B, 64+nc, A) f32 -> boxes (B,A,4) xyxy input pixels, scores
    (B,A,nc).  docs/YOLO_TensorRT_Technical.md:72-77."""
    B, _, A = raw.shape
    anchors, strides = make_anchors(size)
    box = raw[:, :4 * REG_MAX].view(B, 4, REG_MAX, A).permute(0, 1, 3, 2)
    dist = box.softmax(-1) @ torch.arange(REG_MAX, dtype=torch.float32)      # (B,4,A)
    a = anchors.t()[None]                                                     # (1,2,A)
    x1y1 = a - dist[:, :2]
    x2y2 = a + dist[:, 2:]
    boxes = torch.cat([x1y1, x2y2], 1) * strides[None, None]
    return boxes.transpose(1, 2).contiguous(), raw[:, 4 * REG_MAX:].sigmoid().transpose(1, 2).contiguous()


def forward_raw(sd: Dict[str, torch.Tensor], x: torch.Tensor, scale: str = "n", nc: int = 5,
                return_feats: bool = False):
    """x (B,3,S,S) f32 in [0,1] -> raw head output (B, 64+nc, A)."""
    outs: Dict[int, torch.Tensor] = {}
    for idx, kind, a in topology(scale):
        p = f"model.{idx}."
        if kind == "conv":
            x = _conv(sd, p + "conv", x, a[2], a[3])
        elif kind == "c2f":
            x = _c2f(sd, p, x, a[2], a[3])
        elif kind == "sppf":
            y = [_conv(sd, p + "cv1.conv", x, 1, 1)]
            for _ in range(3):
                y.append(F.max_pool2d(y[-1], 5, 1, 2))
            x = _conv(sd, p + "cv2.conv", torch.cat(y, 1), 1, 1)
        elif kind == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        elif kind == "cat":
            x = torch.cat([x, outs[a[0]]], 1)
        elif kind == "detect":
            feats = [outs[15], outs[18], x]
            res = []
            for s, f in enumerate(feats):
                b = _conv(sd, p + f"cv2.{s}.1.conv", _conv(sd, p + f"cv2.{s}.0.conv", f, 3, 1), 3, 1)
                b = _conv(sd, p + f"cv2.{s}.2", b, 1, 1, act=False)
                c = _conv(sd, p + f"cv3.{s}.1.conv", _conv(sd, p + f"cv3.{s}.0.conv", f, 3, 1), 3, 1)
                c = _conv(sd, p + f"cv3.{s}.2", c, 1, 1, act=False)
                res.append(torch.cat([b, c], 1).flatten(2))
            x = torch.cat(res, 2)
        outs[idx] = x
    return (x, outs) if return_feats else x


def blob(images_u8_nhwc: torch.Tensor) -> torch.Tensor:
    """A2 (解读.md:70-74): RGB HWC u8 -> CHW f32 /255."""
    return images_u8_nhwc.permute(0, 3, 1, 2).to(torch.float32) / 255.0
