"""Drop-in call surface of the reference's `utils.trainYolo` (utils/trainYolo.py:6-35,124-137).

`train(epochs, batch, data)` keeps the reference's signature and runs the MI355X detector training step
(yvhip.yolo_training.YoloTrainer: un-fused YOLOv8 forward with BatchNorm batch statistics, v8 detection loss,
backward, SGD) over the YOLO-format dataset the `data` yaml names.  What `ultralytics` adds around that step is NOT
built and is reported by `train()` in its result: the pre-training `model.val` mAP pass, mosaic / HSV / flip
augmentation, EMA, warm-up, the AdamW choice of `optimizer='auto'`; the pickled `/app/utils/weight/best.pt` cannot be
read with a safe loader, so initial weights come from `weights=` (a state dict written by this trainer) or from a
seeded random initialisation.  Parity unpinned: every piece lives in `ultralytics`, absent from the reference tree.
"""
import json
import os

import torch

import yvhip

from .class_config import xml2txt          # noqa: F401  (same import as the reference module)

WEIGHTS_IN = "/app/utils/weight/best.pth"
WEIGHTS_OUT = "/app/utils/new_weight/yolo_best.pth"
NOT_BUILT = ["model.val (mAP) before training", "mosaic/HSV/flip augmentation", "EMA", "warm-up", "AdamW (optimizer='auto')"]


def train(epochs, batch, data, weights=None, scale="n", size=640, save=None, device="cuda:0", seed=42, log=print):
    """utils/trainYolo.py:6-35: `model.train(epochs=, batch=, data=, lr0=1e-4, lrf=1e-4)`.
    Returns {"epochs": [...per-epoch mean (total, box, cls, dfl)...], "weights": path or None, "not_built": [...]}."""
    from yvhip.yolo_data import list_samples, load_batch, max_boxes_per_image, read_data_yaml
    from yvhip.yolo_training import YoloTrainer, init_yolo_train_state
    yvhip.require_gpu()
    cfg = read_data_yaml(data)
    nc = cfg["nc"]
    samples = list_samples(cfg["train"])
    if not samples:
        raise yvhip.YvError(f"no training images under {cfg['train']}")
    B = max(1, min(int(batch), len(samples)))
    weights = weights if weights is not None else (WEIGHTS_IN if os.path.exists(WEIGHTS_IN) else None)
    if weights is not None:
        state = torch.load(weights, map_location="cpu", weights_only=True)
    else:
        log(f"trainYolo.train: no readable initial weights, seeded random initialisation (seed {seed})")
        state = init_yolo_train_state(scale, nc, seed)
    lr0, lrf = 1e-4, 1e-4
    # ultralytics: weight decay scaled by batch * accumulate / 64 with accumulate = max(round(64 / batch), 1)
    wd = 5e-4 * B * max(round(64 / B), 1) / 64
    tr = YoloTrainer(state, scale=scale, nc=nc, size=size, batch=B, lr=lr0, momentum=0.937, weight_decay=wd, device=device)
    G = max_boxes_per_image(samples)
    hist = []
    for ep in range(int(epochs)):
        lr = lr0 * ((1 - ep / max(int(epochs), 1)) * (1.0 - lrf) + lrf)            # linear lr0 -> lr0*lrf
        acc, steps = torch.zeros(4), 0
        for i in range(0, len(samples) - B + 1, B):
            img, gtb, gtl, gtn = load_batch(samples[i:i + B], size, G)
            loss = tr.step(img.to(device), gtb.to(device), gtl.to(device), gtn.to(device), lr)
            acc += loss.cpu()
            steps += 1
        mean = (acc / max(steps, 1)).tolist()
        hist.append({"epoch": ep, "lr": lr, "loss": mean[0], "box": mean[1], "cls": mean[2], "dfl": mean[3], "steps": steps})
        log(f"epoch {ep}: loss {mean[0]:.4f} box {mean[1]:.4f} cls {mean[2]:.4f} dfl {mean[3]:.4f} ({steps} steps, lr {lr:.3g})")
    out = save if save is not None else WEIGHTS_OUT
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        torch.save(tr.state_dict(), out)
    except OSError as e:
        log(f"trainYolo.train: cannot write {out}: {e}")
        out = None
    return {"epochs": hist, "weights": out, "not_built": NOT_BUILT}


def yoloRetrain():
    """utils/trainYolo.py:124-137 (app.py:99-100 runs this on a background thread and ignores the result):
    VOC xml -> YOLO txt, then one epoch at batch 1."""
    try:
        print("Converting XML annotations to YOLO TXT format...")
        xml2txt("/app/train/new")
        print("Starting YOLO model retraining...")
        results = train(epochs=1, batch=1, data="/app/train/yolo/config.yaml")
        print("YOLO model training finished. Results:", json.dumps(results["epochs"]))
    except Exception as e:
        print(f"yoloRetrain: {e}")
        return False
    return True
