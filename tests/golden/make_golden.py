#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's
own Python functions (build container only: /root/reference is mounted here and
does not exist on the GPU box).  Nothing from the reference is copied: the
outputs are data (inputs + expected outputs).

The reference modules import third-party packages that are absent from this
image (cv2, timm, albumentations, oss2, flask_sse, ultralytics); they are only
*named* at import time on the functions used here, so empty placeholder
modules are registered for them.  `custom_nms` exists only as a code block in
README.md (lines 62-84): the block text is extracted at run time and exec'd
with a local `box_iou` (torchvision's published formula; torchvision absent).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import io
import json
import os
import random
import re
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _placeholders():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    mod("cv2", INTER_NEAREST=0)
    mod("timm")
    mod("albumentations")
    mod("oss2")
    mod("flask_sse", sse=object())
    mod("ultralytics", YOLO=object)
    try:
        import requests  # noqa: F401
    except Exception:
        mod("requests")


def box_iou(b1, b2):
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


def readme_custom_nms():
    txt = open(os.path.join(REF, "README.md"), encoding="utf-8").read()
    m = re.search(r"```python\n(def custom_nms.*?)```", txt, re.S)
    ns = {"torch": torch, "box_iou": box_iou}
    exec(m.group(1), ns)
    return ns["custom_nms"]


sys.path.insert(0, OUT)
from inputs import coord_image, seeded_boxes  # noqa: E402  (shared with tests/)


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    _placeholders()
    from PIL import Image
    import utils.class_config as cc
    import utils.trainClass as tc
    import utils.utils as uu

    gold = {}

    # ---- G1 / G2: crop_image ------------------------------------------------
    cases = []
    tmp = tempfile.mkdtemp()
    sizes = [(640, 480), (640, 640), (1920, 1080), (37, 29), (300, 257)]
    rng = random.Random(7)
    for (W, H) in sizes:
        p = os.path.join(tmp, f"c_{W}x{H}.png")
        Image.fromarray(coord_image(W, H)).save(p)
        boxes = [(100, 50, 300, 250), (0, 0, 37, 29), (W - 40, H - 40, W, H), (5, 5, 16, 14),
                 (0, 0, W, H), (-7, -3, 50, 60), (W - 10, H - 10, W + 25, H + 31), (3, 4, 4, 5),
                 (10, 10, 29, 30), (10, 10, 30, 29)]
        for _ in range(12):
            x0 = rng.randint(-20, W - 2); y0 = rng.randint(-20, H - 2)
            boxes.append((x0, y0, x0 + rng.randint(1, W), y0 + rng.randint(1, H)))
        for b in boxes:
            if b[2] > W + 40 or b[3] > H + 40:
                b = (b[0], b[1], min(b[2], W + 40), min(b[3], H + 40))
            try:
                im = tc.crop_image(p, *b, training=False)
            except ValueError:          # PIL refuses right<left / lower<upper: the reference raises
                cases.append({"W": W, "H": H, "box": list(b), "error": "ValueError"})
                continue
            arr = np.array(im)
            if arr.size == 0:           # zero-area crop: nothing to resize downstream
                cases.append({"W": W, "H": H, "box": list(b), "size": list(im.size), "empty": True})
                continue
            r, g, bl = [int(v) for v in arr[0, 0]]
            ox = r + 256 * (bl % 16)
            oy = g + 256 * (bl // 16)
            cases.append({"W": W, "H": H, "box": list(b), "origin": [ox, oy], "size": list(im.size)})
    gold["G1_crop_eval"] = cases

    tcases = []
    p = os.path.join(tmp, "c_640x480.png")
    for seed in (0, 1, 2, 3, 4):
        for b in [(100, 50, 300, 250), (5, 5, 16, 14), (600, 440, 640, 480), (0, 0, 320, 200)]:
            random.seed(seed)
            im = tc.crop_image(p, *b, training=True)
            arr = np.array(im)
            r, g, bl = [int(v) for v in arr[0, 0]]
            tcases.append({"W": 640, "H": 480, "seed": seed, "box": list(b),
                           "origin": [r + 256 * (bl % 16), g + 256 * (bl // 16)], "size": list(im.size)})
    gold["G2_crop_train"] = tcases

    # ---- G3: losses ---------------------------------------------------------
    lcases = []
    for B, seed in ((1, 0), (4, 0), (4, 1), (32, 2), (256, 3)):
        g = torch.Generator().manual_seed(seed)
        x = (torch.randn(B, 5, generator=g) * 2).requires_grad_(True)
        lab = torch.randint(0, 5, (B,), generator=g)
        if B == 4 and seed == 0:
            x = torch.manual_seed(0) and torch.randn(4, 5).requires_grad_(True)
            lab = torch.tensor([0, 3, 4, 1])
        y = torch.nn.functional.one_hot(lab, 5).float()
        ls = tc.LabelSmoothingCrossEntropy(0.1)(x, y)
        fo = tc.FocalLoss()(x, y)
        tot = tc.build_loss(x, y)
        tot.backward()
        lcases.append({"x": x.detach().tolist(), "label": lab.tolist(), "lsce": float(ls), "focal": float(fo),
                       "total": float(tot), "grad": x.grad.tolist()})
    gold["G3_loss"] = lcases

    # ---- G4: cosine LR ------------------------------------------------------
    gold["G4_lr"] = {str(E): [tc.cosine_anneal_schedule(t, E, 1e-4) for t in range(2 * E + 1)] for E in (1, 10, 7)}

    # ---- G5: getCorrect -----------------------------------------------------
    g = torch.Generator().manual_seed(5)
    out = torch.randn(16, 5, generator=g)
    lab = torch.randint(0, 5, (16,), generator=g)
    eq, cm = tc.getCorrect(out, torch.nn.functional.one_hot(lab, 5).float())
    gold["G5_correct"] = {"out": out.tolist(), "label": lab.tolist(), "eq": eq.int().tolist(), "cm": cm.tolist()}

    # ---- G6: Network_Wrapper head ------------------------------------------
    class FakeBackbone(torch.nn.Module):
        def forward(self, x):
            return x
    torch.manual_seed(6)
    net = uu.Network_Wrapper(FakeBackbone(), 5)
    net2 = tc.Network_Wrapper(FakeBackbone(), 5)
    net2.load_state_dict(net.state_dict())
    feats = torch.randn(8, 1000)
    y1 = net(feats); y2 = net2(feats)
    assert torch.equal(y1, y2)
    np.savez(os.path.join(OUT, "G6_wrapper.npz"), feats=feats.numpy(), out=y1.detach().numpy(),
             **{k.replace(".", "__"): v.numpy() for k, v in net.state_dict().items()})
    gold["G6_keys"] = list(net.state_dict().keys())
    import inspect
    gold["G6_build_model_sig"] = {"utils.utils": str(inspect.signature(uu.build_model)),
                                  "utils.trainClass": str(inspect.signature(tc.build_model))}

    # ---- G7: custom_nms -----------------------------------------------------
    cn = readme_custom_nms()
    ncases = []
    for n, seed in ((0, 0), (1, 0), (2, 0), (2, 1), (3, 2), (17, 3), (100, 4), (100, 5), (1000, 6), (8400, 7)):
        for thr in (0.45, 0.65):
            if n == 0:
                b, s = torch.zeros(0, 4), torch.zeros(0)
            else:
                b, s = seeded_boxes(n, seed)
            keep = cn(b, s, thr)
            ncases.append({"n": n, "seed": seed, "thr": thr, "keep": [int(k) for k in keep]})
    # IoU == thr exactly: boxes [0,0,2,1] vs [1,0,3,1]... inter 1, union 3 -> 1/3; use thr = float32(1/3)
    b = torch.tensor([[0., 0., 2., 1.], [1., 0., 3., 1.], [10., 10., 12., 12.]])
    s = torch.tensor([0.9, 0.8, 0.7])
    thr_eq = float(box_iou(b[:1], b[1:2])[0, 0])
    ncases.append({"explicit_boxes": b.tolist(), "scores": s.tolist(), "thr": thr_eq, "keep": cn(b, s, thr_eq)})
    # identical + nested + degenerate boxes
    b = torch.tensor([[5., 5., 50., 50.], [5., 5., 50., 50.], [10., 10., 20., 20.], [7., 7., 7., 7.],
                      [7., 7., 7., 7.], [100., 100., 180., 160.]])
    s = torch.tensor([0.5, 0.6, 0.9, 0.3, 0.2, 0.1])
    ncases.append({"explicit_boxes": b.tolist(), "scores": s.tolist(), "thr": 0.45, "keep": cn(b, s, 0.45)})
    gold["G7_custom_nms"] = ncases

    # ---- G8: class_config.convert ------------------------------------------
    gold["G8_convert"] = [{"box": bx, "wh": wh, "out": list(cc.convert(bx, *wh))}
                          for bx, wh in (((10, 20, 110, 220), (640, 480)), ((0, 0, 1, 1), (3, 7)),
                                         ((5, 9, 333, 444), (1920, 1080)))]

    # ---- G9: generate_annotation -------------------------------------------
    d = tempfile.mkdtemp()
    objs = [{"sort": "good", "xmin": 1, "ymin": 2, "xmax": 30, "ymax": 40},
            {"sort": "loss", "xmin": 5, "ymin": 6, "xmax": 70, "ymax": 80},
            {"sort": 4, "xmin": 0, "ymin": 0, "xmax": 9, "ymax": 9},
            {"sort": "weird", "xmin": 3, "ymin": 3, "xmax": 4, "ymax": 4}]
    buf = io.StringIO()
    so = sys.stdout; sys.stdout = buf
    try:
        path = uu.generate_annotation("image", "a_b.jpg", "/app/image/a_b.jpg", objs, save_dir=d + "/")
    finally:
        sys.stdout = so
    gold["G9_annotation"] = {"objects": objs, "xml": open(path, encoding="utf-8").read(),
                             "name": os.path.basename(path)}

    # ---- CFG ---------------------------------------------------------------
    gold["CFG"] = {k: (str(v) if k == "device" else v) for k, v in vars(cc.CFG).items() if not k.startswith("_")}

    with open(os.path.join(OUT, "golden.json"), "w", encoding="utf-8") as f:
        json.dump(gold, f, indent=1, ensure_ascii=False)
    print("wrote", os.path.join(OUT, "golden.json"), {k: (len(v) if hasattr(v, "__len__") else v) for k, v in gold.items()})


if __name__ == "__main__":
    main()
