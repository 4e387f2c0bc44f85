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


# ---------------------------------------------------------------------------------------------------------
# VOC XML -> YOLO txt dataset conversion (utils/class_config.py:44-154): on-disk formats
# either side of the hot path.  File I/O only, no device work.
# ---------------------------------------------------------------------------------------------------------
import os
import random
import shutil
import xml.etree.ElementTree as ET

LABEL_IDS = {'good': 0, 'broke': 1, 'lose': 2, 'loss': 2, 'uncovered': 3, 'circle': 4}
YOLO_ROOT = "/app/train/yolo/fold0"


def copy_image(source_path, destination_folder):
    os.makedirs(destination_folder, exist_ok=True)
    shutil.copy(source_path, os.path.join(destination_folder, os.path.basename(source_path)))


def mkdir(number):
    """./fold{number}/{images,labels}/{train,val}"""
    fold = f"./fold{number}"
    for kind in ("images", "labels"):
        for split in ("train", "val"):
            os.makedirs(os.path.join(fold, kind, split), exist_ok=True)


def writeTxt(path, objects, line_end="\\n"):
    """One line per object: "{label} {cx:.5f} {cy:.5f} {w:.5f} {h:.5f}" + line_end into `{path}.txt`.
    The reference terminates lines with the two characters backslash + n (utils/class_config.py:84), so its
    label files are a single physical line; that byte layout is the default here for format compatibility.
    Pass line_end="\n" for files other YOLO tooling can read."""
    with open(f"{path}.txt", 'w') as f:
        for box in objects['objects']:
            x, y, w, h = convert((box['xmin'], box['ymin'], box['xmax'], box['ymax']), objects["width"], objects["height"])
            f.write("{} {:.5f} {:.5f} {:.5f} {:.5f}{}".format(box["label"], x, y, w, h, line_end))


def _label_of(text):
    if text in LABEL_IDS:
        return LABEL_IDS[text]
    return int(text)                     # annotations written by generate_annotation hold the numeric id


def parse_voc_dir(directory):
    """[{path, objects:[{name,label,xmin,ymin,xmax,ymax}], width, height, name}] for every *.xml in `directory`."""
    from PIL import Image
    out = []
    for file in sorted(os.listdir(directory)):
        if not file.endswith(".xml"):
            continue
        xml_path = os.path.join(directory, file)
        root = ET.parse(xml_path).getroot()
        data_path = os.path.normpath(os.path.join(os.path.dirname(xml_path), root.find('path').text))
        w = int(root.find('size/width').text or 0)
        h = int(root.find('size/height').text or 0)
        if not (w and h):
            with Image.open(data_path) as img:
                w, h = img.size
        objs = []
        for obj in root.findall('.//object'):
            tag = obj.find('name')
            if tag is None:
                tag = obj.find('sort')
            objs.append({'name': tag.text, 'label': _label_of(tag.text),
                         'xmin': int(obj.find('.//xmin').text), 'ymin': int(obj.find('.//ymin').text),
                         'xmax': int(obj.find('.//xmax').text), 'ymax': int(obj.find('.//ymax').text)})
        out.append({'path': data_path, 'objects': objs, 'width': w, 'height': h,
                    'name': os.path.splitext(root.find('filename').text)[0]})
    return out


def xml2pd(directory, yolo_root=YOLO_ROOT, line_end="\\n"):
    """Random 80/20 train/val split (one random.random() draw per image, in directory order), image copy and
    label file per image (utils/class_config.py:89-148)."""
    for item in parse_voc_dir(directory):
        split = "train" if random.random() > 0.2 else "val"
        copy_image(item["path"], os.path.join(yolo_root, "images", split))
        os.makedirs(os.path.join(yolo_root, "labels", split), exist_ok=True)
        writeTxt(os.path.join(yolo_root, "labels", split, item["name"]), item, line_end)


def xml2txt(path, yolo_root=YOLO_ROOT):
    xml2pd(path, yolo_root)
