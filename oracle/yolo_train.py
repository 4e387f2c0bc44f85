"""Oracle (TEST INFRASTRUCTURE): fp32 CPU restatement of the detector TRAINING step, row C4 of SURVEY.md
section 8(a) (`YOLO(pt).train(epochs, batch, data, lr0=1e-4, lrf=1e-4)`, utils/trainYolo.py:13-35).

Parity UNPINNED: model, loss, assigner and optimiser all live inside `ultralytics` (not in requirements.txt,
absent from the reference tree and from this image; no fixture in the tree pins any of them).  This file restates
the published ultralytics v8 definitions:
  * un-fused modules: Conv = Conv2d(bias=False) -> BatchNorm2d(eps 1e-3, momentum 0.03) -> SiLU; C2f / SPPF / Detect
    as docs/YOLO_TensorRT_Technical.md:160-212 describes the fused ones; Detect's last 1x1 convs are plain Conv2d
    with bias;
  * v8DetectionLoss: TaskAlignedAssigner(topk 10, alpha 0.5, beta 6.0), CIoU box loss, DFL (reg_max 16), BCE class
    loss, gains box 7.5 / cls 0.5 / dfl 1.5, total multiplied by the batch size.
Gradients come from torch autograd; the HIP path is compared with them in tests/test_gpu_yolo_train.py."""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from .yolo import REG_MAX, conv_shapes, make_anchors, topology

BN_EPS, BN_MOMENTUM = 1e-3, 0.03


def block_keys(scale: str = "n", nc: int = 5) -> List[Tuple[str, int, int, int, int, bool]]:
    """(key, cin, cout, k, stride, has_bn): `key + '.conv.weight'` / `key + '.bn.*'` for Conv blocks,
    `key + '.weight'` / `key + '.bias'` for Detect's final convs (ultralytics un-fused naming)."""
    out = []
    for key, ci, co, k, s in conv_shapes(scale, nc):
        if key.endswith(".conv"):
            out.append((key[:-5], ci, co, k, s, True))
        else:
            out.append((key, ci, co, k, s, False))
    return out


def init_train_state(scale: str = "n", nc: int = 5, seed: int = 7) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for key, ci, co, k, _, bn in block_keys(scale, nc):
        w = torch.randn(co, ci, k, k, generator=g) * math.sqrt(2.0 / (ci * k * k))
        if bn:
            sd[key + ".conv.weight"] = w
            sd[key + ".bn.weight"] = 1 + 0.1 * torch.randn(co, generator=g)
            sd[key + ".bn.bias"] = 0.1 * torch.randn(co, generator=g)
            sd[key + ".bn.running_mean"] = torch.zeros(co)
            sd[key + ".bn.running_var"] = torch.ones(co)
        else:
            sd[key + ".weight"] = w
            sd[key + ".bias"] = 0.1 * torch.randn(co, generator=g)
    return sd


_EMULATE_BF16 = False       # forward_train(emulate_bf16=True): store z and activations as the device path does


def _q(t: torch.Tensor) -> torch.Tensor:
    """bf16 storage with a straight-through gradient (rounding has zero derivative almost everywhere)."""
    if not _EMULATE_BF16:
        return t
    return t + (t.detach().to(torch.bfloat16).float() - t.detach())


def _cbs(sd, key, x, k, s, train=True, res=None):
    y = _q(F.conv2d(x, sd[key + ".conv.weight"], None, stride=s, padding=k // 2))
    y = F.batch_norm(y, sd[key + ".bn.running_mean"], sd[key + ".bn.running_var"], sd[key + ".bn.weight"],
                     sd[key + ".bn.bias"], training=train, momentum=BN_MOMENTUM, eps=BN_EPS)
    y = _q(F.silu(y))
    return y if res is None else _q(y + res)


def _c2f(sd, p, x, n, shortcut, train):
    y = list(_cbs(sd, p + "cv1", x, 1, 1, train).chunk(2, 1))
    for j in range(n):
        t = _cbs(sd, p + f"m.{j}.cv1", y[-1], 3, 1, train)
        y.append(_cbs(sd, p + f"m.{j}.cv2", t, 3, 1, train, res=y[-1] if shortcut else None))
    return _cbs(sd, p + "cv2", torch.cat(y, 1), 1, 1, train)


def forward_train(sd: Dict[str, torch.Tensor], x: torch.Tensor, scale: str = "n", nc: int = 5, train: bool = True,
                  return_feats: bool = False, emulate_bf16: bool = False):
    """x (B,3,S,S) f32 in [0,1] -> per scale (box logits (B,64,h,w), class logits (B,nc,h,w)).
    Running statistics in `sd` are updated in place when train=True (as nn.BatchNorm2d does).
    emulate_bf16: round every pre-activation and activation to bf16 where the device path stores them (fp32 math in
    between, straight-through gradients) - random-init BatchNorm stacks amplify that storage noise to ~10 % at the
    head, so tests compare the device against THIS variant tightly and against pure fp32 loosely."""
    global _EMULATE_BF16
    _EMULATE_BF16 = emulate_bf16
    try:
        return _forward_train(sd, x, scale, nc, train, return_feats)
    finally:
        _EMULATE_BF16 = False


def _forward_train(sd, x, scale, nc, train, return_feats):
    outs: Dict[int, torch.Tensor] = {}
    res = []
    for idx, kind, a in topology(scale):
        p = f"model.{idx}."
        if kind == "conv":
            x = _cbs(sd, f"model.{idx}", x, a[2], a[3], train)
        elif kind == "c2f":
            x = _c2f(sd, p, x, a[2], a[3], train)
        elif kind == "sppf":
            y = [_cbs(sd, p + "cv1", x, 1, 1, train)]
            for _ in range(3):
                y.append(F.max_pool2d(y[-1], 5, 1, 2))
            x = _cbs(sd, p + "cv2", torch.cat(y, 1), 1, 1, train)
        elif kind == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        elif kind == "cat":
            x = torch.cat([x, outs[a[0]]], 1)
        elif kind == "detect":
            for s, f in enumerate((outs[15], outs[18], x)):
                b = _cbs(sd, p + f"cv2.{s}.1", _cbs(sd, p + f"cv2.{s}.0", f, 3, 1, train), 3, 1, train)
                b = F.conv2d(b, sd[p + f"cv2.{s}.2.weight"], sd[p + f"cv2.{s}.2.bias"])
                c = _cbs(sd, p + f"cv3.{s}.1", _cbs(sd, p + f"cv3.{s}.0", f, 3, 1, train), 3, 1, train)
                c = F.conv2d(c, sd[p + f"cv3.{s}.2.weight"], sd[p + f"cv3.{s}.2.bias"])
                res.append((b, c))
        outs[idx] = x
    return (res, outs) if return_feats else res


def run_module(sd, idx: int, x: torch.Tensor, scale: str = "n", train: bool = True, emulate_bf16: bool = False):
    """One backbone / neck module (conv, c2f, sppf) on its own input (for c2f modules of the neck: the concatenated
    input).  Used by the local-consistency test: device inputs in, device output gradients back."""
    global _EMULATE_BF16
    _EMULATE_BF16 = emulate_bf16
    try:
        kind, a = {i: (k, a) for i, k, a in topology(scale)}[idx]
        p = f"model.{idx}."
        if kind == "conv":
            return _cbs(sd, f"model.{idx}", x, a[2], a[3], train)
        if kind == "c2f":
            return _c2f(sd, p, x, a[2], a[3], train)
        if kind == "sppf":
            y = [_cbs(sd, p + "cv1", x, 1, 1, train)]
            for _ in range(3):
                y.append(F.max_pool2d(y[-1], 5, 1, 2))
            return _cbs(sd, p + "cv2", torch.cat(y, 1), 1, 1, train)
        raise ValueError(kind)
    finally:
        _EMULATE_BF16 = False


def run_detect_scale(sd, s: int, f: torch.Tensor, train: bool = True, emulate_bf16: bool = False):
    """Detect branch of scale s on feature map f -> (box logits (B,64,h,w), class logits (B,nc,h,w))."""
    global _EMULATE_BF16
    _EMULATE_BF16 = emulate_bf16
    try:
        p = "model.22."
        b = _cbs(sd, p + f"cv2.{s}.1", _cbs(sd, p + f"cv2.{s}.0", f, 3, 1, train), 3, 1, train)
        b = F.conv2d(b, sd[p + f"cv2.{s}.2.weight"], sd[p + f"cv2.{s}.2.bias"])
        c = _cbs(sd, p + f"cv3.{s}.1", _cbs(sd, p + f"cv3.{s}.0", f, 3, 1, train), 3, 1, train)
        c = F.conv2d(c, sd[p + f"cv3.{s}.2.weight"], sd[p + f"cv3.{s}.2.bias"])
        return b, c
    finally:
        _EMULATE_BF16 = False


# ------------------------------------------------------------------------------------------ v8 detection loss
def bbox_ciou(b1: torch.Tensor, b2: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """CIoU of xyxy boxes (..., 4) -> (...); the aspect term's alpha is a constant (no gradient), as published."""
    x1, y1, x2, y2 = b1.unbind(-1)
    X1, Y1, X2, Y2 = b2.unbind(-1)
    w1, h1 = x2 - x1, y2 - y1 + eps
    w2, h2 = X2 - X1, Y2 - Y1 + eps
    inter = (torch.min(x2, X2) - torch.max(x1, X1)).clamp(0) * (torch.min(y2, Y2) - torch.max(y1, Y1)).clamp(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.max(x2, X2) - torch.min(x1, X1)
    ch = torch.max(y2, Y2) - torch.min(y1, Y1)
    c2 = cw * cw + ch * ch + eps
    rho2 = ((X1 + X2 - x1 - x2) ** 2 + (Y1 + Y2 - y1 - y2) ** 2) / 4
    v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)) ** 2
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def assign(pd_scores, pd_boxes, anchors, gt_labels, gt_boxes, mask_gt, topk=10, alpha=0.5, beta=6.0, eps=1e-9):
    """TaskAlignedAssigner.  pd_scores (B,A,nc) sigmoid scores, pd_boxes (B,A,4) xyxy in input pixels, anchors (A,2)
    pixel centres, gt_labels (B,G) int64, gt_boxes (B,G,4) xyxy pixels, mask_gt (B,G) bool.
    Returns target_boxes (B,A,4), target_scores (B,A,nc), fg (B,A) bool, target_gt_idx (B,A)."""
    B, A, nc = pd_scores.shape
    G = gt_boxes.shape[1]
    lt = anchors[None, None] - gt_boxes[..., None, :2]                     # (B,G,A,2)
    rb = gt_boxes[..., None, 2:] - anchors[None, None]
    in_gts = torch.cat([lt, rb], -1).amin(-1) > eps                       # anchor centre strictly inside the box
    mask = in_gts & mask_gt[..., None]
    ind = gt_labels.clamp(0, nc - 1)
    cls_scores = pd_scores.permute(0, 2, 1).gather(1, ind[..., None].expand(B, G, A))      # (B,G,A)
    ious = bbox_ciou(gt_boxes[:, :, None].expand(B, G, A, 4), pd_boxes[:, None].expand(B, G, A, 4)).clamp(0)
    ious = torch.where(mask, ious, torch.zeros_like(ious))
    cls_scores = torch.where(mask, cls_scores, torch.zeros_like(cls_scores))
    align = cls_scores.pow(alpha) * ious.pow(beta)
    # top-k anchors per ground truth by the alignment metric (inside the box only)
    # build-defined tie rule (torch.topk leaves it open): candidates are the anchors inside the box, equal metrics
    # (e.g. 0 when the predicted box does not overlap) are taken in ascending anchor order
    metric = torch.where(in_gts, align, torch.full_like(align, -1.0))
    k = min(topk, A)
    idx = metric.sort(dim=-1, descending=True, stable=True).indices[..., :k]
    top = torch.zeros_like(metric, dtype=torch.int32)
    top.scatter_add_(-1, idx, torch.ones_like(idx, dtype=torch.int32))
    mask_pos = (top > 0) & mask
    # an anchor claimed by several ground truths goes to the one with the highest CIoU
    cnt = mask_pos.sum(1)                                                 # (B,A)
    multi = cnt > 1
    best = ious.argmax(1)                                                 # (B,A)
    onehot = F.one_hot(best, G).permute(0, 2, 1).bool()
    mask_pos = torch.where(multi[:, None], onehot, mask_pos)
    fg = mask_pos.any(1)
    tgt = mask_pos.float().argmax(1)                                      # (B,A)
    tb = gt_boxes.gather(1, tgt[..., None].expand(B, A, 4))
    tl = gt_labels.gather(1, tgt).clamp(0)
    ts = F.one_hot(tl, nc).float() * fg[..., None]
    # normalise: score = align * (max CIoU of that gt) / (max align of that gt)
    align = align * mask_pos
    pos_align = align.amax(-1, keepdim=True)
    pos_iou = (ious * mask_pos).amax(-1, keepdim=True)
    norm = (align * pos_iou / (pos_align + eps)).amax(1)                  # (B,A)
    return tb, ts * norm[..., None], fg, tgt


def detection_loss(outs, gt_labels, gt_boxes, mask_gt, size: int, nc: int, gains=(7.5, 0.5, 1.5)):
    """outs: per scale (box logits (B,64,h,w), class logits (B,nc,h,w)); ground truth boxes xyxy in input pixels.
    Returns (total * B, (box, cls, dfl))."""
    B = outs[0][0].shape[0]
    box = torch.cat([b.flatten(2) for b, _ in outs], 2).permute(0, 2, 1)            # (B,A,64)
    cls = torch.cat([c.flatten(2) for _, c in outs], 2).permute(0, 2, 1)            # (B,A,nc)
    anchors, strides = make_anchors(size)                                           # grid units (+0.5), stride
    A = anchors.shape[0]
    proj = torch.arange(REG_MAX, dtype=torch.float32)
    dist = box.view(B, A, 4, REG_MAX).softmax(-1) @ proj                            # (B,A,4) ltrb in grid units
    pb = torch.cat([anchors[None] - dist[..., :2], anchors[None] + dist[..., 2:]], -1)      # xyxy grid units
    with torch.no_grad():
        tb, ts, fg, _ = assign(cls.sigmoid(), pb * strides[None, :, None], anchors * strides[:, None], gt_labels,
                               gt_boxes, mask_gt)
    tss = ts.sum().clamp(min=1.0)
    l_cls = F.binary_cross_entropy_with_logits(cls, ts, reduction="none").sum() / tss
    l_box = torch.zeros(())
    l_dfl = torch.zeros(())
    if fg.any():
        tbg = tb / strides[None, :, None]
        w = ts.sum(-1)[fg]
        iou = bbox_ciou(pb[fg], tbg[fg])
        l_box = ((1.0 - iou) * w).sum() / tss
        a = anchors[None].expand(B, A, 2)[fg]
        t = torch.cat([a - tbg[fg][:, :2], tbg[fg][:, 2:] - a], -1).clamp(0, REG_MAX - 1 - 0.01)
        tl = t.long()
        tr = tl + 1
        wl = tr - t
        wr = 1 - wl
        lg = box.view(B, A, 4, REG_MAX)[fg].view(-1, REG_MAX)
        d = (F.cross_entropy(lg, tl.view(-1), reduction="none").view(tl.shape) * wl +
             F.cross_entropy(lg, tr.view(-1), reduction="none").view(tl.shape) * wr).mean(-1)
        l_dfl = (d * w).sum() / tss
    total = gains[0] * l_box + gains[1] * l_cls + gains[2] * l_dfl
    return total * B, (l_box.detach(), l_cls.detach(), l_dfl.detach())
