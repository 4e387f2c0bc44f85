"""Drop-in for the reference's `utils.class_config` (call surface only; hot-path config object).

`CFG` carries the same attributes and values as utils/class_config.py:12-24 (the configured classifier
is the patch-8 / 785-token ViT; BASELINE.json's benchmark configs use the patch-16 model - both run on
the HIP path).
"""
import torch


class CFG:
    seed = 42
    device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    img_size = [224, 224]
    train_bs = 1
    valid_bs = train_bs * 2
    num_classes = 5
    epoch = 10
    lr = 1e-4
    modelName = "vit_base_patch8_224.augreg_in21k"
    pretrained = '/app/utils/weight/best.pth'
    train_path = ["/app/train/new_train", "/app/train/circle", "/app/train/2024/train_xmls", "/app/train/new"]
    valid_path = ["/app/train/2024/valid_xmls", "/app/train/new_valid"]


def convert(box, dw, dh):
    """(xmin, ymin, xmax, ymax) -> YOLO (cx, cy, w, h) normalised by image width dw / height dh
    (utils/class_config.py:28-42)."""
    cx = (box[0] + box[2]) / 2.0
    cy = (box[1] + box[3]) / 2.0
    w = box[2] - box[0]
    h = box[3] - box[1]
    return cx / dw, cy / dh, w / dw, h / dh
