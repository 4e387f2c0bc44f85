"""`model.val(data=, imgsz=640, batch=16, conf=0.25, iou=0.6)` of utils/trainYolo.py:21-26 on the MI355X path:
folded-BatchNorm inference engine + class-aware NMS kernel (yv_efficient_nms: candidates score > conf, suppression
IoU > iou, 300 detections per image) over the `val` split, then the published ultralytics detection metrics on the
host (numpy bookkeeping, as ultralytics does it): IoU matching at 0.50:0.05:0.95, per-class AP by 101-point
interpolation of the precision envelope, mAP50, mAP50-95, precision / recall at the best mean-F1 confidence.
Parity unpinned (ultralytics absent); the metric code is checked on hand-computable cases in tests/test_host_cpu.py."""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

IOUV = np.linspace(0.5, 0.95, 10)


def box_iou_np(a: np.ndarray, b: np.ndarray, eps: float = 1e-7) -> np.ndarray:
    """(n,4) x (m,4) xyxy -> (n,m)."""
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    inter = np.clip(rb - lt, 0, None).prod(2)
    area = lambda x: (x[:, 2] - x[:, 0]) * (x[:, 3] - x[:, 1])
    return inter / (area(a)[:, None] + area(b)[None] - inter + eps)


def match_predictions(pred_cls: np.ndarray, gt_cls: np.ndarray, iou: np.ndarray) -> np.ndarray:
    """iou (n_gt, n_pred) -> correct (n_pred, 10) bool: at every threshold each ground truth is matched to at most one
    prediction of its class and vice versa, best IoU first."""
    correct = np.zeros((pred_cls.shape[0], IOUV.shape[0]), dtype=bool)
    iou = iou * (gt_cls[:, None] == pred_cls[None])
    for i, thr in enumerate(IOUV):
        m = np.array(np.nonzero(iou >= thr)).T
        if m.shape[0]:
            if m.shape[0] > 1:
                m = m[iou[m[:, 0], m[:, 1]].argsort()[::-1]]
                m = m[np.unique(m[:, 1], return_index=True)[1]]
                m = m[np.unique(m[:, 0], return_index=True)[1]]
            correct[m[:, 1].astype(int), i] = True
    return correct


def compute_ap(recall: np.ndarray, precision: np.ndarray) -> float:
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    y = np.interp(x, mrec, mpre)
    return float(np.sum((y[1:] + y[:-1]) * 0.5 * np.diff(x)))


def _smooth(y: np.ndarray, f: float = 0.05) -> np.ndarray:
    nf = round(len(y) * f * 2) // 2 + 1
    p = np.ones(nf // 2)
    yp = np.concatenate((p * y[0], y, p * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode="valid")


def ap_per_class(tp: np.ndarray, conf: np.ndarray, pred_cls: np.ndarray, target_cls: np.ndarray, eps: float = 1e-16):
    """tp (n_pred, 10) bool, conf (n_pred), pred_cls (n_pred), target_cls (n_gt) ->
    dict(ap (n_cls, 10), classes, p, r (per class at the best mean-F1 confidence), map50, map)."""
    order = np.argsort(-conf, kind="stable")
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes, nt = np.unique(target_cls, return_counts=True)
    ap = np.zeros((classes.shape[0], IOUV.shape[0]))
    px = np.linspace(0, 1, 1000)
    p_curve = np.zeros((classes.shape[0], 1000)); r_curve = np.zeros((classes.shape[0], 1000))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        n_l, n_p = nt[ci], int(sel.sum())
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[sel]).cumsum(0)
        tpc = tp[sel].cumsum(0)
        recall = tpc / (n_l + eps)
        precision = tpc / (tpc + fpc)
        r_curve[ci] = np.interp(-px, -conf[sel], recall[:, 0], left=0)
        p_curve[ci] = np.interp(-px, -conf[sel], precision[:, 0], left=1)
        for j in range(IOUV.shape[0]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])
    f1 = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    best = int(_smooth(f1.mean(0), 0.1).argmax()) if classes.shape[0] else 0
    return {"ap": ap, "classes": classes, "p": p_curve[:, best], "r": r_curve[:, best],
            "map50": float(ap[:, 0].mean()) if ap.size else 0.0, "map": float(ap.mean()) if ap.size else 0.0}


def validate(state: Dict[str, torch.Tensor], samples: List[Tuple[str, str]], scale: str, nc: int, size: int = 640,
             batch: int = 16, conf: float = 0.25, iou: float = 0.6, device: str = "cuda:0", max_det: int = 300) -> Dict:
    """Detection metrics of an (un-fused or fused) ultralytics-layout state dict over YOLO-format samples."""
    from YOLOTensorRT.models import fold_batchnorm
    from . import efficient_nms
    from .engines import YoloEngine
    from .yolo_data import load_batch, max_boxes_per_image
    eng = YoloEngine(fold_batchnorm(state), scale, nc, size, device=device)
    G = max_boxes_per_image(samples) if samples else 1
    tps, confs, pcls, tcls = [], [], [], []
    for i in range(0, len(samples), batch):
        chunk = samples[i:i + batch]
        img, gtb, gtl, gtn = load_batch(chunk, size, G)
        boxes, scores = eng(img.to(device))
        num, ob, osc, ol = efficient_nms(boxes, scores, conf, iou, max_det)
        num, ob, osc, ol = num.cpu().numpy()[:, 0], ob.cpu().numpy(), osc.cpu().numpy(), ol.cpu().numpy()
        for b in range(len(chunk)):
            n, g = int(num[b]), int(gtn[b])
            pb, ps, pl = ob[b, :n], osc[b, :n], ol[b, :n]
            tb, tl = gtb[b, :g].numpy(), gtl[b, :g].numpy()
            tcls.append(tl)
            if n == 0:
                continue
            correct = match_predictions(pl, tl, box_iou_np(tb, pb)) if g else np.zeros((n, IOUV.shape[0]), dtype=bool)
            tps.append(correct); confs.append(ps); pcls.append(pl)
    target_cls = np.concatenate(tcls) if tcls else np.zeros((0,))
    if tps:
        res = ap_per_class(np.concatenate(tps), np.concatenate(confs), np.concatenate(pcls), target_cls)
    else:
        res = ap_per_class(np.zeros((0, 10), bool), np.zeros((0,)), np.zeros((0,)), target_cls)
    return {"images": len(samples), "instances": int(target_cls.shape[0]), "precision": float(res["p"].mean()) if res["p"].size else 0.0,
            "recall": float(res["r"].mean()) if res["r"].size else 0.0, "map50": res["map50"], "map50_95": res["map"],
            "ap_per_class": {int(c): res["ap"][k].tolist() for k, c in enumerate(res["classes"])},
            "conf": conf, "iou": iou, "detections": int(sum(len(c) for c in confs))}
