"""`main` / `draw_image` of the missing `YOLOTensorRT.inferdet` (app.py:16,61,77; test.py:28), rebuilt
batch-first on the device pipeline.  Reconstructed from the sibling walkthrough
(YOLOTensorRT_yolodet_py_解读.md:33-116) and the callers' keyword arguments; semantics the tree cannot pin
are defined here and listed in DESIGN.md (ensemble = mean of logits; crops are RGB; the reported `sort` is
the classifier's class).
"""
import os
from typing import Callable, List, Optional

import numpy as np
import torch

import yvhip
from yvhip.pipeline import DetectClassifyPipeline

from .config import CLASSES, COLORS
from .models.utils import letterbox_geometry, path_to_list

BATCH = 32


def draw_image(image, box, cls):
    """Rectangle + "{CLASS}:1" label (解读.md:35-45) with PIL (OpenCV is not required)."""
    from PIL import Image, ImageDraw
    arr = np.asarray(image)
    im = Image.fromarray(arr[..., ::-1].copy())               # callers hand over BGR
    d = ImageDraw.Draw(im)
    color = COLORS[int(cls) % len(COLORS)][::-1]
    x0, y0, x1, y1 = [int(v) for v in box]
    d.rectangle([x0, y0, x1, y1], outline=color, width=2)
    d.text((x0, max(0, y0 - 12)), f"{CLASSES[int(cls)]}:1", fill=color)
    return np.asarray(im)[..., ::-1].copy()


def _load_rgb(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def _pipeline(Engine, model_list) -> DetectClassifyPipeline:
    vits = [m.engine() for m in model_list]
    return DetectClassifyPipeline(Engine.engine, vits, Engine.score_threshold, Engine.iou_threshold, Engine.topk)


def main(Engine, imgs, device=None, model_list: Optional[list] = None, transform=None, aliyunoss=None,
         func: Optional[Callable] = None):
    """Detect + classify every image under `imgs` (directory, file or list).  Returns a JSON-serialisable dict
    {"output": [{"image": name, "objects": [{"sort", "det_sort", "confidence", "xmin", "ymin", "xmax", "ymax"}]}]}.
    `transform` is accepted for signature compatibility: its valid_test branch (nearest resize to 224 +
    Normalize(.5,.5)) is what the fused crop kernel implements.  `func(folder, filename, path, objects)` is
    called per image like test.py:28 does with generate_annotation."""
    yvhip.require_gpu()
    if not model_list:
        raise yvhip.YvError("model_list is empty")
    dev = torch.device(device) if device is not None else Engine.device
    S = Engine.size
    pipe = _pipeline(Engine, model_list)
    paths = path_to_list(imgs)
    results = []
    for s in range(0, len(paths), BATCH):
        chunk = paths[s:s + BATCH]
        arrs = [_load_rgb(p) for p in chunk]
        B = len(arrs)
        Hc, Wc = max(a.shape[0] for a in arrs), max(a.shape[1] for a in arrs)
        canvas = np.zeros((B, Hc, Wc, 3), dtype=np.uint8)
        geom, ratio, dwdh, wh = [], [], [], []
        for i, a in enumerate(arrs):
            h, w = a.shape[:2]
            canvas[i, :h, :w] = a
            r, (dw, dh), (nw, nh), (left, top) = letterbox_geometry(h, w, (S, S))
            geom.append([w, h, nw, nh, left, top]); ratio.append(r); dwdh += [dw, dh]; wh += [w, h]
        src = torch.from_numpy(canvas).to(dev)
        net_in = yvhip.letterbox(src, torch.tensor(geom, dtype=torch.int32, device=dev), S)
        out = pipe(net_in, torch.tensor(ratio, dtype=torch.float32, device=dev),
                   torch.tensor(dwdh, dtype=torch.float32, device=dev), torch.tensor(wh, dtype=torch.int32, device=dev),
                   src_images=src)
        cnt = out["det_count"].cpu().tolist()
        box, score, dlab = out["det_box"].cpu(), out["det_score"].cpu(), out["det_label"].cpu()
        clist, ctot, clab = out["crop_list"].cpu(), int(out["crop_total"][0]), out["cls_label"].cpu()
        cls_of = {(int(clist[k, 0]), int(clist[k, 5])): int(clab[k]) for k in range(ctot)}
        for i, p in enumerate(chunk):
            objs = []
            for k in range(cnt[i]):
                c = cls_of.get((i, k))
                if c is None:                           # degenerate crop: the reference's PIL crop would raise
                    continue
                x0, y0, x1, y1 = box[i, k].tolist()
                objs.append({"sort": CLASSES[c], "det_sort": CLASSES[int(dlab[i, k]) % len(CLASSES)],
                             "confidence": float(score[i, k]), "xmin": x0, "ymin": y0, "xmax": x1, "ymax": y1})
            name = os.path.basename(p)
            if func is not None:
                func(os.path.basename(os.path.dirname(p)) or "image", name, p, objs)
            if aliyunoss is not None and hasattr(aliyunoss, "put_object_from_file"):
                aliyunoss.put_object_from_file(name, p)
            results.append({"image": name, "objects": objs})
    results.sort(key=lambda r: r["image"])
    return {"output": results}
