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
        self.parts = 2                                    # slices whose counts detect_stage publishes (PipelinedRunner: one per stream)
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

    def detect_stage(self, images: torch.Tensor, ratio: Optional[torch.Tensor] = None,
                     dwdh: Optional[torch.Tensor] = None, img_wh: Optional[torch.Tensor] = None) -> dict:
        """Detector + NMS + restore/filter/dedupe/inflate + batch assembly: everything up to the crop list."""
        B, S = images.shape[0], images.shape[1]
        dev = images.device
        if ratio is None:
            c = self._identity_geometry(B, S, dev)
            ratio, dwdh, img_wh = c["ratio"], c["dwdh"], c["wh"]
        num, bb, sc, lb = self.detect(images)
        post = postprocess_dets(num, bb, sc, lb, ratio, dwdh, img_wh, self.conf, self.dedupe_iou, self.coord_mode,
                                self.max_crops)
        per_img = self.max_crops if self.max_crops > 0 else self.topk
        cap = self.capacity if self.capacity else B * per_img
        # (the counts of the classifier's concurrent slices come out of the same kernel: torch arithmetic on the device-side count
        # was four small dependent launches at the head of every classifier pass)
        crop_list, totals = compact_crops(post["det_count"], post["crop_rect"], post["crop_ok"], cap, parts=self.parts)
        out = dict(post)
        out.update(num_dets=num, bboxes=bb, scores=sc, labels=lb, crop_list=crop_list, crop_total=totals[:1], capacity=cap,
                   crop_part_counts=totals[1:])
        return out

    def classify_stage(self, src: torch.Tensor, det: dict, streams: Optional[Sequence[torch.cuda.Stream]] = None) -> dict:
        """Crop gather + ViT ensemble + wrapper head for the crop list produced by detect_stage.  With `streams`
        (2 HIP streams) the crop list is cut in two halves that run concurrently on independent buffer sets:
        the memory-bound kernels of one half (LayerNorm, attention, residual epilogues) overlap the MFMA-bound
        GEMMs of the other."""
        cap, crop_list, total = det["capacity"], det["crop_list"], det["crop_total"]
        dev = src.device
        v0 = self.vits[0]
        logits = torch.empty((cap, v0.nc), dtype=torch.float32, device=dev)     # rows past the crop count: 0 / -1, written by the
        labels = torch.empty((cap,), dtype=torch.int32, device=dev)             # head kernel of the first ensemble member
        w = 1.0 / len(self.vits)                      # ensemble = mean of logits (defined by this build)
        parts = [(0, cap, total, 0, None)]
        if streams is not None and len(streams) >= 2 and cap >= len(streams):
            k = len(streams)
            base, parts, lo = (cap + k - 1) // k, [], 0
            for i, st in enumerate(streams):                           # contiguous slices of the crop list
                n = min(base, cap - lo)
                if n <= 0:
                    break
                pc = det.get("crop_part_counts")
                if pc is not None and pc.numel() == k:
                    cnt = pc[i:i + 1]                                   # device-side scalar written by yv_compact_crops_split
                else:
                    cnt = torch.clamp(total - lo, min=0, max=n)        # (another slicing than detect_stage prepared)
                parts.append((lo, n, cnt, i, st))
                lo += n
        cur = torch.cuda.current_stream()
        for lo, n, cnt, slot, st in parts:
            ctx = torch.cuda.stream(st) if st is not None else _Null()
            if st is not None:
                st.wait_stream(cur)
            with ctx, v0.guard(slot):                     # the patch buffer belongs to ensemble member 0's buffer set
                patches = crop_resize_norm(src, crop_list[lo:lo + n], cnt, n, v0.img, v0.P, layout=2,
                                           out=v0.patch_buffer(n, slot))
                for i, v in enumerate(self.vits):
                    with v.guard(slot):                     # features live in engine-owned buffers until head() has read them
                        feats = v.backbone(patches, n, cnt, slot)
                        v.head(feats, n, logits[lo:lo + n], labels[lo:lo + n], scale=w, accumulate=i > 0, count=cnt)
        for _, _, _, _, st in parts:
            if st is not None:
                cur.wait_stream(st)
        det = dict(det)
        det.update(cls_logits=logits, cls_label=labels)
        return det

    def __call__(self, images: torch.Tensor, ratio: Optional[torch.Tensor] = None,
                 dwdh: Optional[torch.Tensor] = None, img_wh: Optional[torch.Tensor] = None,
                 src_images: Optional[torch.Tensor] = None) -> dict:
        """images (B,S,S,3) u8 letterboxed network input; src_images (B,H,W,3) u8 originals the crops are
        taken from (default: `images`, i.e. inputs that are already S x S: ratio 1, dwdh 0)."""
        det = self.detect_stage(images, ratio, dwdh, img_wh)
        return self.classify_stage(images if src_images is None else src_images, det)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class PipelinedRunner:
    """Two-stream software pipeline over consecutive batches: while the classifier (large GEMMs that fill the
    chip) works on batch i on one HIP stream, the detector of batch i+1 - dozens of small, latency-bound
    kernels that leave most CUs idle - runs on a second stream in the gaps.  Every batch still passes through
    every stage; only the schedule changes (throughput mode).  `submit()` returns the batch's result dict,
    valid once `sync()` (or the dict's "done" event) has completed.

    Hazards: the detector's activation buffers are private to the detect stream (in order); the classifier's
    buffers are private to the classify stream; the hand-off (crop list, counts, detections) is a fresh allocation
    per batch on the detect stream, published by an event and `record_stream`-ed for the classify stream - torch's
    caching allocator would otherwise hand a dropped batch's blocks to the NEXT detect pass while the classifier of
    that batch has not read them yet (seen as a memory fault with YOLOv8m + ViT-L at batch 64).  Back-pressure: the
    detect stream may run at most one batch ahead of the classify stream (it waits for the classifier of batch i-2).
    The detect stream is created with HIGH priority: HIP maps equal-priority streams round-robin onto a few hardware
    queues, and two streams that land on the same queue do not overlap at all (measured in one process: 9.58 ms for an
    unlucky pair vs 8.61 ms with the priority split; a lucky pair gives the same 8.6-8.9 ms)."""

    def __init__(self, pipe: DetectClassifyPipeline, split_classifier=False, run_ahead: int = 1,
                 det_priority: int = -1, gemm_cus: Optional[int] = 208, full_cus_from: Optional[int] = None):
        """gemm_cus: workgroups of the persistent classifier GEMMs (yv_set_option "linear_p8_cus": per thread, set around this
        runner's own classifier submissions and restored afterwards).  Those workgroups own a CU each for a whole launch (160 KB of LDS, 256 VGPRs x 8 waves), so with all
        256 CUs taken the kernels of the other streams (detector, the other half-batch's LayerNorm / attention) can only
        start when a GEMM ends; leaving 48 CUs free lets them run alongside: 8.07 -> 7.75 ms per step measured (208 and 200
        equal, 224 7.85, 192 and below worse again).  None keeps the library default (every CU)."""
        self.pipe = pipe
        self.gemm_cus = None if gemm_cus is None else int(gemm_cus)
        self.full_cus_from = full_cus_from                         # transformer block from which the GEMMs take every CU again
        self.run_ahead = max(int(run_ahead), 1)                    # batches the detect stream may lead the classifier by
        self.s_det = torch.cuda.Stream(priority=det_priority)
        self.s_cls = torch.cuda.Stream()
        n_split = 2 if split_classifier is True else int(split_classifier or 0)      # True = 2 parts, or the number of parts
        from . import HW_QUEUES
        if n_split >= 2 and HW_QUEUES < 8:
            n_split = 0          # with HIP's default 4 hardware queues the extra streams collide and the split is slower
        self.s_sub = [torch.cuda.Stream() for _ in range(n_split)] if n_split >= 2 else None
        if n_split >= 2:
            pipe.parts = n_split                                   # detect_stage publishes one crop count per classifier stream
        self._last = None
        self._done = []                                            # "classifier finished" events of the last two batches

    def submit(self, images: torch.Tensor, ratio=None, dwdh=None, img_wh=None, src_images=None) -> dict:
        cur = torch.cuda.current_stream()
        self.s_det.wait_stream(cur)                               # inputs produced on the caller's stream
        # the caller may drop its input tensors right after submit(): tell the caching allocator which streams still read
        # them (same hazard as the hand-off tensors below; bench.py reuses one tensor and never hit it)
        for t in (images, ratio, dwdh, img_wh, src_images):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                for st in [self.s_det, self.s_cls] + list(self.s_sub or []):
                    t.record_stream(st)
        if len(self._done) > self.run_ahead:
            self.s_det.wait_event(self._done[-1 - self.run_ahead])  # at most `run_ahead` batches ahead of the classifier
        with torch.cuda.stream(self.s_det):
            det = self.pipe.detect_stage(images, ratio, dwdh, img_wh)
            for t in det.values():
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(self.s_cls)                     # consumed on the classify stream
                    for st in (self.s_sub or []):
                        t.record_stream(st)
            ready = torch.cuda.Event()
            ready.record(self.s_det)
        with torch.cuda.stream(self.s_cls):
            self.s_cls.wait_event(ready)
            from . import get_option, set_option
            prev = get_option("linear_p8_cus")
            if self.gemm_cus is not None:
                set_option("linear_p8_cus", self.gemm_cus)
            for v in self.pipe.vits:
                v.full_cus_from = self.full_cus_from
            try:
                out = self.pipe.classify_stage(images if src_images is None else src_images, det, self.s_sub)
            finally:
                set_option("linear_p8_cus", prev)
            done = torch.cuda.Event()
            done.record(self.s_cls)
        out["done"] = done
        self._done = (self._done + [done])[-(self.run_ahead + 1):]
        self._last = out
        return out

    def sync(self):
        self.s_det.synchronize()
        self.s_cls.synchronize()
        for st in (self.s_sub or []):
            st.synchronize()
