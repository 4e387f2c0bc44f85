"""Oracle (TEST INFRASTRUCTURE): the hot path arranged like the reference - a batch-1
per-image loop on the CPU in fp32.  Used by the
end-to-end parity test, by __graft_entry__.smoke() and as bench.py's `cpu_baseline`
("port").  Never imported by the product."""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import boxes as ob
from . import vit as ov
from . import yolo as oy


def post_stages(num: int, bb, sc, lb, ratio: float, dwdh, wh, conf=0.35, dedupe_iou=0.45, coord_mode="trunc",
                max_crops=0):
    """restore -> filter -> int -> custom_nms -> inflate for ONE image.  Returns list of dicts."""
    idx, ib, s, l = ob.restore_and_filter(num, bb, sc, lb, ratio, dwdh, conf, coord_mode)
    n = int(num)
    d = torch.tensor([dwdh[0], dwdh[1], dwdh[0], dwdh[1]], dtype=torch.float32)
    fb = (bb[:n].float() - d) / torch.tensor(ratio, dtype=torch.float32)
    sel = torch.tensor(idx, dtype=torch.long)
    keep = ob.custom_nms(fb[sel], sc[:n][sel], dedupe_iou) if len(idx) else []
    if max_crops > 0:
        keep = keep[:max_crops]
    dets = []
    for k in keep:
        rect = ob.inflate_eval(*ib[k], wh[0], wh[1])
        dets.append(dict(box=ib[k], score=s[k], label=l[k], rect=list(rect),
                         ok=bool(rect[2] > rect[0] and rect[3] > rect[1])))
    return dets


def run_image(img_u8_hwc: torch.Tensor, yolo_sd, vit_sds, vit_name: str, scale="n", nc=5, max_crops=0,
              coord_mode="trunc") -> Dict:
    """One S x S RGB u8 image through the whole path (identity letterbox)."""
    S = img_u8_hwc.shape[0]
    raw = oy.forward_raw(yolo_sd, oy.blob(img_u8_hwc[None]), scale, nc)
    boxes, scores = oy.decode(raw, nc, S)
    num, bb, sc, lb = ob.efficient_nms(boxes, scores)
    dets = post_stages(num[0, 0], bb[0], sc[0], lb[0], 1.0, (0.0, 0.0), (S, S), max_crops=max_crops,
                       coord_mode=coord_mode)
    crops = [ob.crop_resize_normalize(img_u8_hwc.numpy(), d["rect"]) for d in dets if d["ok"]]
    if crops:
        x = torch.from_numpy(np.stack(crops))
        logits = sum(ov.wrapper_forward(sd, x, vit_name) for sd in vit_sds) / len(vit_sds)
        labels = logits.argmax(1).tolist()
    else:
        logits, labels = torch.zeros(0, nc), []
    return dict(dets=dets, logits=logits, labels=labels)
