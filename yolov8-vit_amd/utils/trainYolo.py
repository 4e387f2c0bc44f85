"""Drop-in call surface of the reference's `utils.trainYolo` (utils/trainYolo.py:6-35,124-137).

`train(epochs, batch, data)` keeps the reference's signature and runs the MI355X detector training step
(yvhip.yolo_training.YoloTrainer: un-fused YOLOv8 forward with BatchNorm batch statistics, v8 detection loss,
backward, SGD) over the YOLO-format dataset the `data` yaml names.  Around that step the published trainer's defaults
are followed: Mosaic -> RandomPerspective(scale, translate) -> HSV -> flip augmentation composed on the device
(yvhip/yolo_augment.py, mosaic closed for the last 10 epochs), `optimizer='auto'` (AdamW / Nesterov SGD), warm-up,
nominal-batch accumulation, ModelEMA, validation after every epoch with best-fitness checkpoint selection, `val()` before
and after.  The pickled `/app/utils/weight/best.pt` cannot be read with a safe loader, so initial weights come from `weights=` (a state dict written by this trainer) or from a
seeded random initialisation.  Parity unpinned: every piece lives in `ultralytics`, absent from the reference tree.
"""
import json
import os
import random

import torch

import yvhip

from .class_config import xml2txt          # noqa: F401  (same import as the reference module)

WEIGHTS_IN = "/app/utils/weight/best.pth"
WEIGHTS_OUT = "/app/utils/new_weight/yolo_best.pth"
NOT_BUILT = ["rotation / shear / perspective / mixup / copy-paste augmentation (all 0 in the default configuration)"]
NBS = 64                                   # ultralytics nominal batch size


def _auto_optimizer(n_images, batch, epochs, nc, lr0, momentum):
    """ultralytics `optimizer='auto'`: more than 10000 iterations -> SGD(lr0 0.01, nesterov), else AdamW with
    lr0 = round(0.002 * 5 / (4 + nc), 6); momentum 0.9 either way and the caller's lr0 / momentum are ignored."""
    iterations = -(-n_images // max(batch, NBS)) * epochs
    if iterations > 10000:
        return "sgd_nesterov", 0.01, 0.9
    return "adamw", round(0.002 * 5 / (4 + nc), 6), 0.9


def val(state, data, scale="n", imgsz=640, batch=16, conf=0.25, iou=0.6, device="cuda:0"):
    """`model.val(data=, imgsz=640, batch=16, conf=0.25, iou=0.6, device='0')` (utils/trainYolo.py:21-26): detection
    metrics of a state dict over the yaml's `val` split (yvhip/yolo_val.py)."""
    from yvhip.yolo_data import list_samples, read_data_yaml
    from yvhip.yolo_val import validate
    from yvhip.yolo_augment import DetAugment, augment_batch
    cfg = read_data_yaml(data)
    samples = list_samples(cfg["val"]) if cfg["val"] and os.path.isdir(cfg["val"]) else []
    return validate(state, samples, scale, cfg["nc"], imgsz, batch, conf, iou, device)


def train(epochs, batch, data, weights=None, scale="n", size=640, save=None, device="cuda:0", seed=42, log=print,
          optimizer="auto", lr0=1e-4, lrf=1e-4, momentum=0.937, weight_decay=5e-4, warmup_epochs=3.0, augment=True,
          close_mosaic=10):
    """utils/trainYolo.py:6-35: `model.train(epochs=, batch=, data=, lr0=1e-4, lrf=1e-4)` with the ultralytics defaults
    around it: optimizer 'auto' (see _auto_optimizer), linear lr0 -> lr0*lrf schedule, warm-up over
    max(3 epochs, 100 iterations) (lr from 0 - biases from 0.1, 0.0 under AdamW - and momentum from 0.8), gradient
    accumulation to the nominal batch 64 with weight decay scaled by batch*accumulate/64, ModelEMA(0.9999, tau 2000)
    whose weights are validated after every epoch (conf 0.001, IoU 0.7); the epoch with the best fitness
    (0.1 mAP50 + 0.9 mAP50-95) is saved to `save` (best) next to `<save>_last` (last).
    Returns {"epochs": [...per-epoch mean (total, box, cls, dfl)...], "weights": path or None, "not_built": [...]}."""
    import numpy as np
    from yvhip.yolo_data import list_samples, load_batch, max_boxes_per_image, read_data_yaml
    from yvhip.yolo_val import validate
    from yvhip.yolo_augment import DetAugment, augment_batch
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
    validation_results = val(state, data, scale, size, 16, 0.25, 0.6, device)
    log(f"Validation results before training: {validation_results}")
    opt, warmup_bias_lr = optimizer.lower(), 0.1
    if opt == "auto":
        opt, lr0, momentum = _auto_optimizer(len(samples), B, int(epochs), nc, lr0, momentum)
        warmup_bias_lr = 0.0 if opt == "adamw" else 0.1
        log(f"trainYolo.train: optimizer=auto -> {opt}, lr0={lr0}, momentum={momentum} (lr0/momentum arguments ignored, as ultralytics does)")
    elif opt == "sgd":
        opt = "sgd_nesterov"                                   # ultralytics builds SGD with nesterov=True
    accumulate = max(round(NBS / B), 1)
    wd = weight_decay * B * accumulate / NBS
    tr = YoloTrainer(state, scale=scale, nc=nc, size=size, batch=B, lr=lr0, momentum=momentum, weight_decay=wd, device=device,
                     optimizer=opt, ema=True)
    G = max_boxes_per_image(samples)
    nb = max(len(samples) // B, 1)                             # batches per epoch (last short batch dropped)
    nw = max(round(warmup_epochs * nb), 100) if warmup_epochs > 0 else -1
    lf = lambda ep: max(1 - ep / max(int(epochs), 1), 0) * (1.0 - lrf) + lrf
    hist, ni = [], 0
    val_samples = list_samples(cfg["val"]) if cfg["val"] and os.path.isdir(cfg["val"]) else []
    best_fit, best_state, best_epoch = None, None, -1
    order_rng = random.Random(seed)
    det_aug, mosaic_on, G_aug = (DetAugment(size, seed=seed) if augment else None), True, 4 * G
    for ep in range(int(epochs)):
        lr = lr0 * lf(ep)
        acc, steps = torch.zeros(4), 0
        order = list(samples)
        order_rng.shuffle(order)                               # the trainer's loader shuffles every epoch
        if ep == int(epochs) - int(close_mosaic):              # the trainer closes the mosaic for the last epochs
            mosaic_on = False
        for i in range(0, len(order) - B + 1, B):
            lrs, mom, acc_now = None, None, accumulate
            if ni <= nw:                                       # warm-up (ultralytics trainer, per iteration)
                xi = [0, nw]
                acc_now = max(1, int(np.interp(ni, xi, [1, NBS / B]).round()))
                lrs = {g: float(np.interp(ni, xi, [warmup_bias_lr if g == "bias" else 0.0, lr])) for g in ("w", "bnw", "bias")}
                mom = float(np.interp(ni, xi, [0.8, momentum]))
            if det_aug is not None:                            # Mosaic -> affine -> HSV -> flip, composed on the device
                img, gtb, gtl, gtn = augment_batch(order, range(i, i + B), det_aug, G_aug, device, use_mosaic=mosaic_on)
            else:
                img, gtb, gtl, gtn = load_batch(order[i:i + B], size, G)
            loss = tr.step(img.to(device), gtb.to(device), gtl.to(device), gtn.to(device), lr, accumulate=acc_now, lrs=lrs,
                           momentum=mom)
            acc += loss.cpu()
            steps += 1
            ni += 1
        mean = (acc / max(steps, 1)).tolist()
        row = {"epoch": ep, "lr": lr, "loss": mean[0], "box": mean[1], "cls": mean[2], "dfl": mean[3], "steps": steps}
        ema_state = tr.state_dict(ema=True)
        if val_samples:
            # the trainer validates the EMA weights after every epoch (conf 0.001, NMS IoU 0.7) and keeps the checkpoint
            # with the best fitness = 0.1 * mAP50 + 0.9 * mAP50-95
            m = validate(ema_state, val_samples, scale, nc, size, 16, 0.001, 0.7, device)
            fit = 0.1 * m["map50"] + 0.9 * m["map50_95"]
            row.update(map50=m["map50"], map50_95=m["map50_95"], fitness=fit)
            if not best_fit or best_fit < fit:
                best_fit = fit
            if best_fit == fit:
                best_state, best_epoch = {k: v.detach().cpu().clone() for k, v in ema_state.items()}, ep
        hist.append(row)
        log(f"epoch {ep}: loss {mean[0]:.4f} box {mean[1]:.4f} cls {mean[2]:.4f} dfl {mean[3]:.4f} ({steps} steps, lr {lr:.3g})"
            + (f" mAP50 {row['map50']:.4f} mAP50-95 {row['map50_95']:.4f}" if val_samples else ""))
    last_state = tr.state_dict(ema=True)
    if best_state is None:                                     # no validation split: the last EMA weights are the result
        best_state, best_epoch = last_state, int(epochs) - 1
    out = save if save is not None else WEIGHTS_OUT
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        torch.save(best_state, out)                            # best.pt
        root, ext = os.path.splitext(out)
        torch.save(last_state, root + "_last" + ext)           # last.pt
    except OSError as e:
        log(f"trainYolo.train: cannot write {out}: {e}")
        out = None
    final = val(best_state, data, scale, size, 16, 0.25, 0.6, device)
    log(f"Validation results after training: {final}")
    return {"epochs": hist, "weights": out, "not_built": NOT_BUILT, "optimizer": opt, "lr0": lr0, "accumulate": accumulate,
            "val_before": validation_results, "val_after": final, "best_epoch": best_epoch, "best_fitness": best_fit}


def yolo2dict(path_to_xml_dir):
    """utils/trainYolo.py:41-122: VOC xml directory -> [(image file name, [{'name': class id, 'xmin', 'ymin', 'xmax',
    'ymax'}, ...])] sorted by file name; class from <name> (falling back to <sort>), numeric strings '0'..'4' taken as
    ids, unknown names -> -1; image name = xml stem + '.jpg' ('.png' for the one file the reference special-cases)."""
    import xml.etree.ElementTree as ET
    label_mapping = {'good': 0, 'broke': 1, 'lose': 2, 'loss': 2, 'uncovered': 3, 'circle': 4}
    results = []
    for xml_file in [f for f in os.listdir(path_to_xml_dir) if f.endswith('.xml')]:
        base = os.path.splitext(xml_file)[0]
        objs = []
        for obj in ET.parse(os.path.join(path_to_xml_dir, xml_file)).getroot().findall('object'):
            tag = obj.find('name') if obj.find('name') is not None else obj.find('sort')
            name = tag.text
            label = int(name) if name in ['0', '1', '2', '3', '4'] else label_mapping.get(name, -1)
            bb = obj.find('bndbox')
            objs.append({'name': label, 'xmin': int(bb.find('xmin').text), 'ymin': int(bb.find('ymin').text),
                         'xmax': int(bb.find('xmax').text), 'ymax': int(bb.find('ymax').text)})
        results.append((base + ('.png' if base == "test152" else '.jpg'), objs))
    results.sort(key=lambda x: x[0])
    return results


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
