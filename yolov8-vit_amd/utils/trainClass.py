"""Drop-in for the reference's `utils.trainClass` (names, argument meaning and error style).

Hot-path pieces run on the HIP kernels: `build_model`/`Network_Wrapper` (ViT engine), `build_loss` +
`LabelSmoothingCrossEntropy` + `FocalLoss` (fused loss kernel with its analytic gradient),
`getCorrect` (device argmax), the eval transform and `crop_image`'s integer inflate.  `train_one_epoch` / `valid_one_epoch`
run the native fine-tune step (yvhip.training).  The dataset front-end (`xml2pd`, `build_dataset`, `build_dataloader`,
`deliver`, `train`, `retrain`) follows the reference on the host; the stochastic training transforms
run on the device (yvhip/augment.py, csrc/augment.hip); ONNX export is not rebuilt.
"""
import json
import math
import os
import random
import time

import numpy as np
import torch
import torch.nn as nn
from PIL import Image

import yvhip
from yvhip.modules import Network_Wrapper, build_network  # noqa: F401

from .class_config import *  # noqa: F401,F403  (CFG, like the reference module)
from .class_config import CFG


# ------------------------------------------------------------------------------------------ losses
class _FusedLoss(torch.autograd.Function):
    """loss = w_ls*LSCE(0.1) + w_fo*Focal(1,2,'mean'); forward and d/dlogits come from one kernel."""

    @staticmethod
    def forward(ctx, x, onehot, w_ls, w_fo):
        yvhip.require_gpu()
        dev = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
        logits = x.detach().to(dev, torch.float32).contiguous()
        labels = onehot.detach().to(dev).argmax(1).to(torch.int32).contiguous()
        loss, grad = yvhip.loss_fwd_bwd(logits, labels, w_ls, w_fo)
        ctx.save_for_backward(grad)
        ctx.src = x
        return loss[0].to(x.device)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g.to(grad.device)).to(ctx.src.device, ctx.src.dtype), None, None, None


class FocalLoss(nn.Module):
    """utils/trainClass.py:46-66 (alpha 1, gamma 2, reduction 'mean' are the only values the path uses)."""

    def __init__(self, alpha=1, gamma=2, reduction='mean'):
        super().__init__()
        if alpha != 1 or gamma != 2 or reduction != 'mean':
            raise yvhip.YvError("the fused loss kernel implements FocalLoss(alpha=1, gamma=2, reduction='mean')")

    def forward(self, inputs, targets):
        return _FusedLoss.apply(inputs, targets, 0.0, 1.0)


class LabelSmoothingCrossEntropy(nn.Module):
    """utils/trainClass.py:162-185 with smoothing 0.1 (the value build_loss uses)."""

    def __init__(self, smoothing=0.1):
        super().__init__()
        assert 0.0 < smoothing < 1.0
        if abs(smoothing - 0.1) > 1e-12:
            raise yvhip.YvError("the fused loss kernel implements smoothing = 0.1")
        self.smoothing, self.confidence = smoothing, 1.0 - smoothing

    def forward(self, x, targets):
        return _FusedLoss.apply(x, targets, 1.0, 0.0)


def build_loss(x, y):
    """utils/trainClass.py:362-370: LSCE(0.1)/6 + Focal*5/6 on logits x and one-hot y."""
    return _FusedLoss.apply(x, y, 1.0 / 6.0, 5.0 / 6.0)


# ------------------------------------------------------------------------------- crops / transforms
def inflate_box(x_min, y_min, x_max, y_max, width, height, training=False):
    """Integer inflate + clamp of crop_image (utils/trainClass.py:76-91); random draw order
    x_max, x_min, y_max, y_min in the training branch."""
    dis_x = (x_max - x_min) // 10
    dis_y = (y_max - y_min) // 10
    if training:
        x_max = min(width, x_max + random.randint(0, dis_x))
        x_min = max(0, x_min - random.randint(0, dis_x))
        y_max = min(height, y_max + random.randint(0, dis_y))
        y_min = max(0, y_min - random.randint(0, dis_y))
    else:
        x_max = min(width, x_max + dis_x // 2)
        x_min = max(0, x_min - dis_x // 2)
        y_max = min(height, y_max + dis_y // 2)
        y_min = max(0, y_min - dis_y // 2)
    return x_min, y_min, x_max, y_max


def crop_image(image_path, x_min, y_min, x_max, y_max, training=False):
    """utils/trainClass.py:70-93: host-side PIL crop (file I/O, outside the device path); the batch
    pipeline uses the fused crop kernel with the same integer rule."""
    img = Image.open(image_path).convert('RGB')
    w, h = img.size
    return img.crop(inflate_box(x_min, y_min, x_max, y_max, w, h, training))


class _EvalTransform:
    """A.Compose([Resize(h,w,INTER_NEAREST), Normalize(.5,.5)]) as a callable(image=ndarray)->{"image": ...}
    (utils/trainClass.py:218-221).  Host numpy path for API compatibility; same index/rounding rule as the
    crop kernel."""

    def __init__(self, size):
        self.h, self.w = size

    @staticmethod
    def _idx(dst, src):
        ifx = 1.0 / (float(dst) / float(src))
        return np.minimum(np.floor(np.arange(dst) * ifx).astype(np.int64), src - 1)

    def __call__(self, image):
        a = np.asarray(image)
        g = a[self._idx(self.h, a.shape[0])][:, self._idx(self.w, a.shape[1])]
        x = g.astype(np.float32)
        x -= np.float32(127.5)
        x *= np.reciprocal(np.float32(127.5), dtype=np.float32)
        return {"image": x}


class _TrainTransform(_EvalTransform):
    """`data_transforms['train']` (utils/trainClass.py:199-216).  Called per item on the host it performs the
    deterministic head of the sequence (Resize + Normalize, exactly the eval transform); the stochastic transforms
    (flip, crop+pad, shift-scale-rotate, channel shuffle, grid / elastic distortion, coarse dropout) are drawn per
    batch by `device_augment` and applied on the device by `train_one_epoch` while the batch is turned into the
    patch-embed operand (yv_augment_patchify): the reference applies them to the normalised image as well, so moving
    them behind the collate step changes nothing but where they run."""

    def __init__(self, size):
        super().__init__(size)
        from yvhip.augment import TrainAugment
        if size[0] != size[1]:
            raise yvhip.YvError("the device augmentation needs a square CFG.img_size")
        self.device_augment = TrainAugment(size[0])


def build_transforms(CFG):
    """utils/trainClass.py:197-222: {'train': stochastic sequence, 'valid_test': Resize + Normalize}."""
    return {"train": _TrainTransform(CFG.img_size), "valid_test": _EvalTransform(CFG.img_size)}


# -------------------------------------------------------------------------- schedule / accuracy
def cosine_anneal_schedule(t, nb_epoch, lr):
    """utils/trainClass.py:97-105."""
    cos_inner = np.pi * (t % (nb_epoch))
    cos_inner /= (nb_epoch)
    return float(lr / 2 * (np.cos(cos_inner) + 1))


def getCorrect(output_concat, target):
    """utils/trainClass.py:109-117: eq vector (CPU bool tensor) + confusion matrix [true][pred]."""
    predicted = torch.max(output_concat.data, 1).indices
    targets = torch.max(target, 1).indices.int()
    equal = predicted.eq(targets).cpu()
    n = CFG.num_classes
    cm = np.bincount((targets.cpu().numpy().astype(np.int64) * n + predicted.cpu().numpy()), minlength=n * n)
    return equal, cm.reshape(n, n)


def set_seed(seed=42):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


# ---------------------------------------------------------------------------------- model / eval
def build_model(CFG, pretrained=None, modelName=None):
    """utils/trainClass.py:341-358."""
    return build_network(CFG, modelName, pretrained)


def buildInferModel(path="/app/utils/weight/class.onnx"):
    """The reference opens an onnxruntime session on an exported classifier (utils/trainClass.py:546-554)
    and returns None on failure.  The native path needs no export: a wrapper state dict at `path` is loaded
    into the HIP engine; anything else reports the error and returns None."""
    try:
        return build_network(CFG, None, path).eval()
    except Exception as e:
        print(f"buildInferModel: {e}")
        return None


@torch.no_grad()
def valid_one_epoch(net, criterion, testloader):
    """utils/trainClass.py:121-158: mean loss, accuracy in percent, prints the row-normalised confusion matrix."""
    net.eval()
    total_cm = np.zeros((CFG.num_classes, CFG.num_classes), dtype=int)
    test_loss, correct, total, idx = 0.0, 0, 0, 0
    for batch_idx, (inputs, targets, path) in enumerate(testloader):
        idx = batch_idx
        outputs = net(inputs)
        targets = targets.to(outputs.device).float()
        test_loss += float(criterion(outputs, targets))
        eq, cm = getCorrect(outputs.data, targets.data)
        total_cm += cm
        total += targets.size(0)
        correct += int(eq.sum())
        print('Step: %d | Loss: %.3f |Combined Acc: %.3f%% (%d/%d)' % (
            batch_idx, test_loss / (batch_idx + 1), 100. * float(correct) / total, correct, total))
    acc = 100. * float(correct) / max(total, 1)
    with np.errstate(invalid="ignore", divide="ignore"):
        print(total_cm.astype('float') / total_cm.sum(axis=1)[:, np.newaxis])
    return acc, test_loss / (idx + 1)


# ------------------------------------------------------------------------------------- training
def _trainer_for(net, optimizer):
    """One VitTrainer per wrapper module (flat fp32 master weights + momentum live in the trainer)."""
    from yvhip.training import VitTrainer
    tr = getattr(net, "_yv_trainer", None)
    if tr is None:
        mom, wd = 0.9, 1e-3                                   # utils/trainClass.py:442-443
        if optimizer is not None and getattr(optimizer, "param_groups", None):
            g0 = optimizer.param_groups[0]
            mom, wd = g0.get("momentum", mom), g0.get("weight_decay", wd)
        dev = next(net.parameters()).device
        if dev.type != "cuda":
            dev = torch.device("cuda", torch.cuda.current_device())
        sd = {k: v.detach() for k, v in net.state_dict().items()}
        tr = VitTrainer(sd, net.model.arch, net.num_class, net.model.img, device=str(dev), momentum=mom, weight_decay=wd)
        net._yv_trainer = tr
    return tr


def train_one_epoch(net, netp, trainloader, CELoss, optimizer, lr, batch_size, epoch, nb_epoch, use_cuda, device):
    """utils/trainClass.py:374-420 on the HIP path: per batch (short batches skipped, :394) the LR is set from
    `cosine_anneal_schedule(epoch, nb_epoch, lr[0])` (:400-401), then forward / build_loss / backward / SGD run
    as ONE native step (yvhip.training.VitTrainer.step).  `CELoss` must be this module's build_loss (the fused
    kernel implements exactly that loss).  Returns the number of correct predictions like the reference."""
    from yvhip.modules import patchify_bf16
    if CELoss is not build_loss:
        raise yvhip.YvError("the native train step implements build_loss (LSCE/6 + Focal*5/6) only")
    tr = _trainer_for(net, optimizer)
    net.train()
    train_loss, correct, total = 0.0, 0, 0
    # stochastic transforms of data_transforms['train'] run on the device (see _TrainTransform); a loader built with
    # any other transform trains on exactly what it yields
    aug = getattr(getattr(getattr(trainloader, "dataset", None), "transforms", None), "device_augment", None)
    for batch_idx, (inputs, targets, path) in enumerate(trainloader):
        if inputs.shape[0] < batch_size:
            continue
        cur_lr = cosine_anneal_schedule(epoch, nb_epoch, lr[0])
        if optimizer is not None:
            for grp in optimizer.param_groups:
                grp['lr'] = cur_lr
        x = inputs.to(tr.dev).float().contiguous()
        labels = targets.to(tr.dev).argmax(1).to(torch.int32).contiguous()
        if aug is not None:
            geo, idx = aug.sample(x.shape[0])
            patches = yvhip.augment_patchify(x, torch.from_numpy(geo).to(tr.dev), torch.from_numpy(idx).to(tr.dev), tr.P_)
        else:
            patches = patchify_bf16(x, tr.P_)
        loss, logits = tr.step(patches, labels, cur_lr)
        eq, _ = getCorrect(logits.data, targets.to(logits.device).float().data)
        total += targets.size(0)
        correct += int(eq.sum())
        train_loss += float(loss[0])
        print('Step: %d | Loss: %.3f | Acc: %.3f%% (%d/%d)' % (
            batch_idx, train_loss / (batch_idx + 1), 100. * float(correct) / total, correct, total))
    net.load_state_dict(tr.state_dict())                     # so that torch.save(net.state_dict()) sees the update
    return correct


# ------------------------------------------------------------------------------- dataset front-end
LABEL_MAPPING = {'good': 0, 'broke': 1, 'lose': 2, 'loss': 2, 'uncovered': 3, 'circle': 4}      # utils/trainClass.py:278-285
SKIP_IMAGES = ('well5_0011.jpg',)                                                                 # :298


def xml2pd(directory):
    """utils/trainClass.py:277-327: VOC xml files directly inside each listed directory -> (objects, objects_circle),
    one entry per annotated object, both lists shuffled (random.shuffle, objects first).  Class names come from
    <name> (falling back to <sort>); numeric ids written by generate_annotation (<sort>3</sort>) are accepted too."""
    import xml.etree.ElementTree as ET
    objects, objects_circle = [], []
    for d in directory:
        if not os.path.isdir(d):
            continue
        for file in sorted(os.listdir(d)):
            if not file.endswith(".xml"):
                continue
            path = os.path.join(d, file)
            root = ET.parse(path).getroot()
            data_path = os.path.normpath(os.path.join(os.path.dirname(path), root.find('path').text))
            if os.path.basename(data_path) in SKIP_IMAGES:
                continue
            name = os.path.splitext(root.find('filename').text)[0]
            for obj in root.findall('.//object'):
                tag = obj.find('name') if obj.find('name') is not None else obj.find('sort')
                sort = tag.text
                label = LABEL_MAPPING[sort] if sort in LABEL_MAPPING else int(sort)
                temp = {'name': sort, 'label': label, 'xmin': int(obj.find('.//xmin').text), 'ymin': int(obj.find('.//ymin').text),
                        'xmax': int(obj.find('.//xmax').text), 'ymax': int(obj.find('.//ymax').text)}
                (objects_circle if label == 4 else objects).append(
                    {'path': data_path, 'objects': temp, "width": 0, "height": 0, "name": name})
    random.shuffle(objects)
    random.shuffle(objects_circle)
    return objects, objects_circle


class build_dataset(torch.utils.data.Dataset):
    """utils/trainClass.py:227-273: training draws from `objects` or `objects_circle` with the circle share as the
    switch probability (one random.random() per item), validation is the concatenation; crops use crop_image
    (random inflation when training) and the given transform; items are (CHW float tensor, one-hot int64, path)."""

    def __init__(self, objects, objects_circle, val=False, train_val_flag=True, transforms=None):
        self.objects, self.objects_circle = objects, objects_circle
        self.train_val_flag, self.transforms, self.val = train_val_flag, transforms, val
        self.lenth_cir, self.lenth = len(objects_circle), len(objects)
        self.rate = self.lenth_cir / (self.lenth + self.lenth_cir) if (self.lenth + self.lenth_cir) > 0 else 0
        if val:
            self.dataset = objects + objects_circle

    def __len__(self):
        return len(self.objects_circle) + len(self.objects)

    def __getitem__(self, index):
        if not self.val:
            if random.random() > self.rate:
                obj = self.objects[index % self.lenth if self.lenth > 0 else 0]
            else:
                obj = self.objects_circle[index % self.lenth_cir if self.lenth_cir > 0 else 0]
        else:
            obj = self.dataset[index]
        o = obj['objects']
        img = crop_image(obj["path"], o["xmin"], o["ymin"], o["xmax"], o["ymax"], training=not self.val)
        data = self.transforms(image=np.array(img))
        chw = torch.from_numpy(np.ascontiguousarray(np.transpose(data['image'], (2, 0, 1))))
        if not self.train_val_flag:
            return chw, obj["path"]
        label = torch.nn.functional.one_hot(torch.tensor(o["label"]), num_classes=CFG.num_classes)
        return chw, label.to(torch.int64), obj["path"]


def build_dataloader(objects, objects_circle, valid_objects, valid_objects_circle, data_transforms):
    """utils/trainClass.py:331-341 (train: shuffle, drop_last=False; valid: in order)."""
    from torch.utils.data import DataLoader
    train_dataset = build_dataset(objects, objects_circle, val=False, train_val_flag=True, transforms=data_transforms['train'])
    valid_dataset = build_dataset(valid_objects, valid_objects_circle, val=True, train_val_flag=True,
                                  transforms=data_transforms['valid_test'])
    train_loader = DataLoader(train_dataset, batch_size=CFG.train_bs, num_workers=0, shuffle=True, drop_last=False)
    valid_loader = DataLoader(valid_dataset, batch_size=CFG.valid_bs, num_workers=0, shuffle=False)
    return train_loader, valid_loader


def deliver(source_dir="/app/train/new/", dest_dir_train="/app/train/new_train", dest_dir_val="/app/train/new_valid"):
    """utils/trainClass.py:557-595: move image + xml pairs 80/20 (one random.random() per image after a shuffle of the
    listing) into the train / valid folders; images without an xml are skipped with a warning."""
    import shutil
    os.makedirs(dest_dir_train, exist_ok=True)
    os.makedirs(dest_dir_val, exist_ok=True)
    filenames = [f for f in sorted(os.listdir(source_dir)) if f.endswith(('.jpg', '.jpeg', '.png'))]
    random.shuffle(filenames)
    for filename in filenames:
        xml_name = os.path.splitext(filename)[0] + '.xml'
        if not os.path.exists(os.path.join(source_dir, xml_name)):
            print(f"Warning: XML file {os.path.join(source_dir, xml_name)} not found for image {filename}. Skipping.")
            continue
        dest = dest_dir_train if random.random() > 0.2 else dest_dir_val
        try:
            shutil.move(os.path.join(source_dir, filename), os.path.join(dest, filename))
            shutil.move(os.path.join(source_dir, xml_name), os.path.join(dest, xml_name))
        except Exception as e:
            print(f"Error moving file {filename} or its XML: {e}")
    print("Data delivery complete.")


def classExport(CFG, pretrained=None, modelName=None):
    """utils/trainClass.py:512-541 exports the classifier to ONNX for onnxruntime; neither exists on this path (the
    trained state dict is what build_model loads): reported, not raised, like the reference's export errors."""
    print("classExport: ONNX export is not part of the MI355X path; use the saved state dict with utils.utils.build_model")
    return None


def train(CFG, log=False, save_path="/app/utils/new_weight/best.pth"):
    """utils/trainClass.py:424-508: datasets from CFG.train_path / CFG.valid_path, model from CFG.pretrained (random
    init if that file is absent), CFG.epoch epochs of train_one_epoch + valid_one_epoch, best state dict to
    /app/utils/new_weight/best.pth, result.json when `log`.  Training crops get the random inflation of crop_image, the
    deterministic resize + normalise on the host and the stochastic transforms of utils/trainClass.py:199-216 on the
    device (see _TrainTransform)."""
    data_transforms = build_transforms(CFG)
    objects, objects_circle = xml2pd(CFG.train_path)
    valid_objects, valid_objects_circle = xml2pd(CFG.valid_path)
    if not (objects or objects_circle):
        raise yvhip.YvError(f"no annotated objects under {CFG.train_path}")
    train_loader, valid_loader = build_dataloader(objects, objects_circle, valid_objects, valid_objects_circle, data_transforms)
    pretrained = CFG.pretrained if CFG.pretrained and os.path.exists(CFG.pretrained) else None
    if pretrained is None:
        print(f"train: {CFG.pretrained} not found, seeded random initialisation")
    if pretrained is None:
        from yvhip.modules import create_model
        net = Network_Wrapper(create_model(CFG.modelName, pretrained=False, num_classes=1000), CFG.num_classes)
    else:
        net = build_model(CFG, pretrained=pretrained)
    net.to(CFG.device)
    if save_path:
        try:
            os.makedirs(os.path.dirname(save_path), exist_ok=True)
        except OSError as e:
            print(f"train: cannot create {os.path.dirname(save_path)}: {e}")
            save_path = None
    return fit(net, train_loader, valid_loader, CFG, log=log, save_path=save_path)


def fit(net, train_loader, valid_loader, CFG, log=False, save_path=None):
    """Epoch loop of utils/trainClass.py:459-508 for caller-provided loaders: train, validate, keep the best
    state dict (optionally saved), write result.json when `log` (same shape as :475-490)."""
    optimizer = torch.optim.SGD(net.parameters(), CFG.lr, momentum=0.9, weight_decay=1e-3)   # hyper-parameter carrier
    best, results = 0.0, {}
    for epoch_num in range(1, CFG.epoch + 1):
        t0 = time.time()
        train_one_epoch(net, net, train_loader, build_loss, optimizer, [CFG.lr], CFG.train_bs, epoch_num - 1, CFG.epoch,
                        True, CFG.device)
        val_acc, val_loss = valid_one_epoch(net, build_loss, valid_loader)
        results[epoch_num] = {'train_acc': "N/A", 'val_acc': val_acc, 'loss': val_loss}
        if log:
            try:
                with open(log if isinstance(log, str) else '/app/train/result.json', 'w') as f:
                    json.dump(results, f, indent=4)
            except IOError as e:
                print(f"Error: could not write result.json: {e}")
        if val_acc > best:
            best = val_acc
            if save_path:
                torch.save(net.state_dict(), save_path)
        print("epoch:{}, time:{:.2f}s, best_val_acc:{:.2f}%\n".format(epoch_num, time.time() - t0, best), flush=True)
    return results


def retrain(log=False):
    """utils/trainClass.py:598-640 (app.py:91-94,181-184 call this on a background thread and ignore the return value):
    seed, deliver the new images 80/20, clear result.json, train, export.  Reference style: report and return."""
    try:
        set_seed(CFG.seed)
        print("Starting data delivery...")
        deliver()
        if log:
            try:
                with open('/app/train/result.json', 'w') as f:
                    json.dump({}, f)
            except IOError:
                print("Error: Could not clear /app/train/result.json for new training log.")
        print("Starting training...")
        train(CFG, log)
        classExport(CFG, pretrained="/app/utils/new_weight/best.pth")
        print("Retraining process complete.")
    except Exception as e:
        print(f"retrain: {e}")
        return False
    return True
