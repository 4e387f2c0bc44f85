"""Host-side helpers named like the missing `YOLOTensorRT.models.utils` (解读.md:27-30)."""
import os
from pathlib import Path
from typing import List, Tuple, Union

import numpy as np

SUFFIXES = ('.bmp', '.dng', '.jpeg', '.jpg', '.mpo', '.png', '.tif', '.tiff', '.webp', '.pfm')


def path_to_list(images_path: Union[str, Path, list, tuple]) -> List[str]:
    """str (file or directory) / list / tuple -> list of image paths (directory entries filtered by suffix,
    name order)."""
    if isinstance(images_path, (list, tuple)):
        return [str(p) for p in images_path]
    p = Path(images_path)
    if p.is_dir():
        return sorted(str(q) for q in p.iterdir() if q.suffix.lower() in SUFFIXES)
    if p.suffix.lower() not in SUFFIXES:
        raise ValueError(f"{p} is not an image or a directory")
    return [str(p)]


def letterbox_geometry(h: int, w: int, new_shape: Tuple[int, int] = (640, 640)):
    """Published letterbox arithmetic in Python floats: r = min(H/h, W/w); unpad = round(w*r), round(h*r);
    (dw, dh) = half the leftover; top/left = round(d - 0.1).  `new_shape` is (W, H) as at 解读.md:68."""
    W, H = new_shape
    r = min(H / h, W / w)
    nw, nh = int(round(w * r)), int(round(h * r))
    dw, dh = (W - nw) / 2, (H - nh) / 2
    top, left = int(round(dh - 0.1)), int(round(dw - 0.1))
    return r, (dw, dh), (nw, nh), (left, top)


def letterbox(im: np.ndarray, new_shape: Tuple[int, int] = (640, 640), color=(114, 114, 114)):
    """Single-image host version returning (image, ratio, (dw, dh)) like the missing helper; runs the device
    kernel on a batch of one (the batch path in inferdet.main calls yvhip.letterbox directly)."""
    import torch
    import yvhip
    yvhip.require_gpu()
    h, w = im.shape[:2]
    r, dwdh, (nw, nh), (left, top) = letterbox_geometry(h, w, new_shape)
    if new_shape[0] != new_shape[1]:
        raise yvhip.YvError("square network input only")
    src = torch.from_numpy(np.ascontiguousarray(im[..., :3])).cuda()[None]
    geom = torch.tensor([[w, h, nw, nh, left, top]], dtype=torch.int32, device=src.device)
    out = yvhip.letterbox(src.contiguous(), geom, new_shape[0])
    return out[0].cpu().numpy(), r, dwdh


def blob(im: np.ndarray, return_seg: bool = False):
    """HWC u8 -> (1,3,H,W) f32 in [0,1] (解读.md:72-74)."""
    x = np.ascontiguousarray(im.transpose(2, 0, 1)[None]).astype(np.float32) / 255.0
    return (x, None) if return_seg else x
