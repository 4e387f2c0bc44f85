"""Batch-first detect -> NMS -> restore/filter/dedupe -> inflate -> crop -> classify pipeline.

Re-creates, on the device and for a whole batch, what the reference does per image in a
Python loop: Engine() incl. EfficientNMS (tech.md:41-47) -> det_postprocess ->
`bboxes -= dwdh; bboxes /= ratio` -> score >= 0.35 -> int coords (解读.md:82-99) ->
custom_nms dedupe (README.md:41,62-84) -> crop_image inflate (utils/trainClass.py:70-93) ->
transform['valid_test'] (app.py:39-42) -> model_list -> class.  One stream, no host
synchronisation between stages: the crop count stays on the device and every ViT kernel
reads it (rows beyond it exit), so the whole step is graph-capturable.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import (YvError, compact_crops, crop_resize_norm, efficient_nms, postprocess_dets)
from .engines import VitEngine, YoloEngine


class DetectClassifyPipeline:
    def __init__(self, yolo: YoloEngine, vits: Sequence[VitEngine], score_threshold: float = 0.25,
                 iou_threshold: float = 0.65, topk: int = 100, conf: float = 0.35, dedupe_iou: float = 0.45,
                 coord_mode: str = "trunc", max_crops_per_image: int = 0, crops_capacity: Optional[int] = None):
        if not vits:
            raise YvError("model_list is empty")
        p = {(v.P, v.img) for v in vits}
        if len(p) != 1:
            raise YvError("ensemble members must share patch size and input size")
        self.yolo, self.vits = yolo, list(vits)
        self.score_threshold, self.iou_threshold, self.topk = score_threshold, iou_threshold, topk
        self.conf, self.dedupe_iou, self.coord_mode = conf, dedupe_iou, coord_mode
        self.max_crops = max_crops_per_image
        self.capacity = crops_capacity
        self._const: Dict[int, dict] = {}

    def _identity_geometry(self, B: int, S: int, dev):
        key = (B, S)
        if key not in self._const:
            self._const[key] = dict(ratio=torch.ones(B, device=dev), dwdh=torch.zeros(2 * B, device=dev),
                                    wh=torch.full((2 * B,), S, dtype=torch.int32, device=dev))
        return self._const[key]

    def detect(self, images: torch.Tensor):
        boxes, scores = self.yolo(images)
        return efficient_nms(boxes, scores, self.score_threshold, self.iou_threshold, self.topk)

    def __call__(self, images: torch.Tensor, ratio: Optional[torch.Tensor] = None,
                 dwdh: Optional[torch.Tensor] = None, img_wh: Optional[torch.Tensor] = None,
                 src_images: Optional[torch.Tensor] = None) -> dict:
        """images (B,S,S,3) u8 letterboxed network input; src_images (B,H,W,3) u8 originals the crops are
        taken from (default: `images`, i.e. inputs that are already S x S: ratio 1, dwdh 0)."""
        B, S = images.shape[0], images.shape[1]
        dev = images.device
        if ratio is None:
            c = self._identity_geometry(B, S, dev)
            ratio, dwdh, img_wh = c["ratio"], c["dwdh"], c["wh"]
        src = images if src_images is None else src_images
        num, bb, sc, lb = self.detect(images)
        post = postprocess_dets(num, bb, sc, lb, ratio, dwdh, img_wh, self.conf, self.dedupe_iou, self.coord_mode,
                                self.max_crops)
        per_img = self.max_crops if self.max_crops > 0 else self.topk
        cap = self.capacity if self.capacity else B * per_img
        crop_list, total = compact_crops(post["det_count"], post["crop_rect"], post["crop_ok"], cap)
        v0 = self.vits[0]
        patches = crop_resize_norm(src, crop_list, total, cap, v0.img, v0.P, layout=2, out=v0.patch_buffer(cap))
        logits = torch.zeros((cap, v0.nc), dtype=torch.float32, device=dev)
        labels = torch.full((cap,), -1, dtype=torch.int32, device=dev)
        w = 1.0 / len(self.vits)                      # ensemble = mean of logits (defined by this build)
        for i, v in enumerate(self.vits):
            feats = v.backbone(patches, cap, total)
            v.head(feats, cap, logits, labels, scale=w, accumulate=i > 0, count=total)
        out = dict(post)
        out.update(num_dets=num, bboxes=bb, scores=sc, labels=lb, crop_list=crop_list, crop_total=total,
                   cls_logits=logits, cls_label=labels, capacity=cap)
        return out
