"""ctypes binding of libyvhip.so (include/yv_hip.h) + thin tensor-level wrappers.

PyTorch-ROCm is used only as plumbing: device memory (`tensor.data_ptr()`),
the current HIP stream and `torch.distributed`.  Every op below enqueues
hand-written gfx950 kernels through the C ABI; there is NO fallback path: if the
library is missing or a call fails, a `YvError` is raised.
"""
from __future__ import annotations

import ctypes as C
import os

# multi-stream schedules (yvhip.pipeline.PipelinedRunner) need more hardware queues than HIP's default 4; only effective
# when set before the process's first HIP call, harmless otherwise
_HWQ_PRESET = os.environ.get("GPU_MAX_HW_QUEUES")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import re
from typing import List, Optional

import torch

# hardware queues this process will really have: the variable only counts if HIP was not initialised yet
HW_QUEUES = int(_HWQ_PRESET) if _HWQ_PRESET else (4 if torch.cuda.is_initialized() else 8)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libyvhip.so")
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "yv_hip.h"))


class YvError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise YvError(f"{LIB_PATH} is missing: build it with `make -C yolov8-vit_amd/csrc` "
                      "(or __graft_entry__.build()); there is no CPU fallback")
    return C.CDLL(LIB_PATH)


lib = _load()

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class yv_view(C.Structure):
    _fields_ = [("ptr", _vp), ("ld", _i), ("c", _i), ("up", _i)]


_SIGS = {
    "yv_version": (_i, []),
    "yv_error_string": (C.c_char_p, [_i]),
    "yv_device_is_gfx950": (_i, []),
    "yv_set_option": (_i, [C.c_char_p, _i]),
    "yv_get_option": (_i, [C.c_char_p, _vp]),
    "yv_set_workspace": (_i, [_vp, _vp, _sz]),
    "yv_set_launch_timing": (_i, [_vp, _vp]),
    "yv_linear_mxfp8_q": (_i, [_vp, C.c_longlong, _vp, C.c_longlong, _vp, _vp, C.c_longlong, _vp, _i, _i, _i, _i, _vp, _i, _vp,
                               C.c_longlong, _vp, C.c_longlong, _vp]),
    "yv_attention_mxfp8": (_i, [_vp, _i, _i, _i, _f, _vp, C.c_longlong, _vp, C.c_longlong, _vp, _vp]),
    "yv_layernorm_mxfp8": (_i, [_vp, _sz, _vp, _vp, _i, _i, _f, _vp, _sz, _vp, C.c_longlong, _vp, _i, _vp]),
    "yv_mx_probe": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "yv_quant_mxfp8": (_i, [_vp, C.c_longlong, C.c_longlong, _i, _vp, C.c_longlong, _vp, C.c_longlong, _vp]),
    "yv_linear_mxfp8": (_i, [_vp, C.c_longlong, _vp, C.c_longlong, _vp, _vp, C.c_longlong, _vp, _i, _i, _i, _vp, _i, _i, _vp,
                             _i, _vp]),
    "yv_custom_nms_ws_bytes": (_sz, [_i, _i]),
    "yv_custom_nms": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "yv_efficient_nms": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "yv_efficient_nms_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "yv_efficient_nms_ws": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "yv_postprocess_dets": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _f, _f, _i, _i,
                                 _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "yv_compact_crops": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "yv_compact_crops_split": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "yv_crop_resize_norm": (_i, [_vp, _i, _i, _i, _sz, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "yv_letterbox": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "yv_augment_patchify": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "yv_mosaic_augment": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "yv_detect_decode": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "yv_detect_tail": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "yv_c2f_debug": (_i, [_vp]),
    "yv_crop_debug": (_i, [_i]),
    "yv_c2f_fused": (_i, [_vp, C.c_longlong, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_longlong, _vp]),
    "yv_optim_step": (_i, [_i, _vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _f, _i, _vp, _vp]),
    "yv_ema_update": (_i, [_vp, _vp, _sz, _f, _vp]),
    "yv_axpby": (_i, [_vp, _vp, _sz, _f, _f, _vp]),
    "yv_detect_loss_ws_bytes": (_sz, [_i, _i, _i]),
    "yv_detect_loss": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _f, _f, _f, _vp, _vp, _sz, _vp]),
    "yv_blob_nhwc8": (_i, [_vp, C.c_longlong, _vp, _vp]),
    "yv_bn_ws_floats": (_sz, [C.c_longlong, _i]),
    "yv_bn_stats": (_i, [_vp, C.c_longlong, C.c_longlong, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "yv_bn_act_fwd": (_i, [_vp, C.c_longlong, C.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, C.c_longlong, _vp, C.c_longlong,
                           _i, _vp]),
    "yv_bn_act_bwd": (_i, [_vp, C.c_longlong, _vp, C.c_longlong, C.c_longlong, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp,
                           _vp, C.c_longlong, _vp, _sz, _vp]),
    "yv_view_op": (_i, [_i, _vp, C.c_longlong, _vp, C.c_longlong, _i, _i, _i, _i, _vp]),
    "yv_maxpool5_bwd": (_i, [_vp, C.c_longlong, _vp, C.c_longlong, _vp, C.c_longlong, _i, _i, _i, _i, _vp, _sz, _vp]),
    "yv_im2col3": (_i, [_vp, C.c_longlong, _i, _i, _i, _i, _i, _vp, _vp]),
    "yv_conv_weight_dgrad": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "yv_conv2d": (_i, [C.POINTER(yv_view), C.POINTER(yv_view), _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _i,
                       _i, _vp]),
    "yv_conv2d_ws": (_i, [C.POINTER(yv_view), C.POINTER(yv_view), _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _i,
                          _i, _vp, _sz, _vp]),
    "yv_linear": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp]),
    "yv_layernorm": (_i, [_vp, _sz, _vp, _vp, _i, _i, _f, _vp, _sz, _vp, _i, _vp]),
    "yv_attention": (_i, [_vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    "yv_attention_debug": (_i, [_i]),
    "yv_cls_rows": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "yv_wrapper_head": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "yv_sppf_pool": (_i, [_vp, _i, _i, _i, _i, _i, _vp]),
    "yv_stem_conv": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "yv_linear_ex": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _vp]),
    "yv_attention_train": (_i, [_vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    "yv_attention_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    "yv_linear_nn": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _vp]),
    "yv_wgrad": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "yv_wgrad_conv3": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "yv_transpose_bf16": (_i, [_vp, _i, _i, C.c_longlong, _vp, C.c_longlong, _vp]),
    "yv_cast_weights": (_i, [_vp, _i, _i, _vp, _vp, C.c_longlong, _vp]),
    "yv_colsum_ws_floats": (_sz, [_i, _i]),
    "yv_cast_colsum": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "yv_colsum_bf16": (_i, [_vp, _i, _i, C.c_longlong, _vp, _i, _vp, _vp]),
    "yv_layernorm_bwd_ws_floats": (_sz, [_i, _i]),
    "yv_layernorm_bwd": (_i, [_vp, C.c_longlong, _vp, _vp, C.c_longlong, _i, _i, _f, _vp, C.c_longlong, _vp, _vp, _vp, _vp]),
    "yv_token_reduce": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "yv_head_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "yv_loss_fwd_bwd": (_i, [_vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp]),
    "yv_sgd_step": (_i, [_vp, _vp, _vp, _sz, _f, _f, _f, _f, _i, _vp, _vp]),
    "yv_transpose_bf16_batched": (_i, [_vp, _vp, _i, _i, _i, C.c_longlong, C.c_longlong, _vp]),
}


def header_symbols() -> List[str]:
    """Every function include/yv_hip.h declares (used by the CPU symbol test)."""
    txt = open(HEADER_PATH, encoding="utf-8").read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(yv_[a-z0-9_]+)\s*\(", txt)))


MISSING: List[str] = []


def _bind():
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:           # header/library mismatch: recorded, and calling it raises
            MISSING.append(name)
            continue
        fn.restype = res
        fn.argtypes = args


_bind()


def set_option(key: str, value: int):
    check(lib.yv_set_option(key.encode(), int(value)), "yv_set_option")


def get_option(key: str) -> int:
    v = C.c_int(0)
    check(lib.yv_get_option(key.encode(), C.byref(v)), "yv_get_option")
    return int(v.value)


def check(code: int, what: str = ""):
    if code != 0:
        raise YvError(f"{what or 'libyvhip'}: {lib.yv_error_string(code).decode()} ({code})")


def _apply_env_options():
    """YV_OPTIONS="key=value,key=value": tuning knobs of yv_set_option for A/B runs (tools/, bench.py)."""
    spec = os.environ.get("YV_OPTIONS", "")
    for item in filter(None, (x.strip() for x in spec.split(","))):
        k, _, v = item.partition("=")
        set_option(k.strip(), int(v))


_apply_env_options()


def require_gpu():
    if not torch.cuda.is_available():
        raise YvError("no HIP device visible: the hot path has no CPU fallback")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


_STREAM_WS = {}


def _st():
    """Current HIP stream handle; the first use of a stream registers its split-K workspace (64 MB)."""
    h = torch.cuda.current_stream().cuda_stream
    key = (torch.cuda.current_device(), h)
    if key not in _STREAM_WS:
        ws = torch.empty((16 * 1024 * 1024,), dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
        _STREAM_WS[key] = ws
        lib.yv_set_workspace(C.c_void_p(h), C.c_void_p(ws.data_ptr()), ws.numel() * 4)
    return C.c_void_p(h)


def _chk_dev(*ts):
    for t in ts:
        if t is not None:
            if not t.is_cuda:
                raise YvError("expected a device tensor")
            if not t.is_contiguous():
                raise YvError("expected a contiguous tensor")


# ------------------------------------------------------------------ boxes
def custom_nms_batched(boxes: torch.Tensor, scores: torch.Tensor, counts: Optional[torch.Tensor],
                       iou_threshold: float = 0.45):
    """boxes (S,n,4) f32, scores (S,n) f32, counts (S) i32|None -> keep (S,n) i32 (-1 padded), num (S) i32."""
    _chk_dev(boxes, scores, counts)
    S, n = scores.shape
    keep = torch.empty((S, n), dtype=torch.int32, device=boxes.device)
    num = torch.empty((S,), dtype=torch.int32, device=boxes.device)
    wsb = lib.yv_custom_nms_ws_bytes(S, n)
    ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=boxes.device)
    check(lib.yv_custom_nms(_p(boxes), _p(scores), _p(counts), S, n, float(iou_threshold), _p(keep), _p(num),
                            _p(ws), wsb, _st()), "yv_custom_nms")
    return keep, num


def custom_nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float = 0.45) -> List[int]:
    """Reference signature (README.md:62): returns a Python list of kept ORIGINAL indices."""
    require_gpu()
    n = int(scores.shape[0])
    if n == 0:
        return []
    dev = torch.device("cuda", torch.cuda.current_device())
    b = boxes.to(device=dev, dtype=torch.float32).reshape(1, n, 4).contiguous()
    s = scores.to(device=dev, dtype=torch.float32).reshape(1, n).contiguous()
    keep, num = custom_nms_batched(b, s, None, iou_threshold)
    k = int(num[0])
    return keep[0, :k].tolist()


def efficient_nms(boxes: torch.Tensor, scores: torch.Tensor, score_threshold: float = 0.25,
                  iou_threshold: float = 0.65, max_output_boxes: int = 100, pre_nms_topk: int = 4096,
                  single_kernel: bool = False):
    """boxes (B,A,4), scores (B,A,nc) f32 -> (num_dets (B,1) i32, bboxes (B,K,4), scores (B,K), labels (B,K) i32).
    Default: the multi-workgroup form (yv_efficient_nms_ws; scratch from torch's caching allocator, allocated on the current
    stream like the outputs); single_kernel=True runs the one-workgroup-per-image form.  Both give identical outputs."""
    _chk_dev(boxes, scores)
    B, A, nc = scores.shape
    K = max_output_boxes
    dev = boxes.device
    num = torch.empty((B, 1), dtype=torch.int32, device=dev)
    ob = torch.empty((B, K, 4), dtype=torch.float32, device=dev)
    osc = torch.empty((B, K), dtype=torch.float32, device=dev)
    ol = torch.empty((B, K), dtype=torch.int32, device=dev)
    if single_kernel:
        check(lib.yv_efficient_nms(_p(boxes), _p(scores), B, A, nc, float(score_threshold), float(iou_threshold), K,
                                   int(pre_nms_topk), _p(num), _p(ob), _p(osc), _p(ol), _st()), "yv_efficient_nms")
        return num, ob, osc, ol
    wsb = int(lib.yv_efficient_nms_ws_bytes(B, A, nc, K, int(pre_nms_topk)))
    if B > 0 and wsb == 0:
        raise YvError("yv_efficient_nms_ws_bytes: arguments out of range")
    ws = torch.empty((max(wsb, 256),), dtype=torch.uint8, device=dev)
    check(lib.yv_efficient_nms_ws(_p(boxes), _p(scores), B, A, nc, float(score_threshold), float(iou_threshold), K,
                                  int(pre_nms_topk), _p(num), _p(ob), _p(osc), _p(ol), _p(ws), wsb, _st()),
          "yv_efficient_nms_ws")
    return num, ob, osc, ol


def postprocess_dets(num_dets, bboxes, scores, labels, ratio, dwdh, img_wh, conf=0.35, dedupe_iou=0.45,
                     coord_mode: str = "trunc", max_crops: int = 0):
    _chk_dev(num_dets, bboxes, scores, labels, ratio, dwdh, img_wh)
    B, slots = scores.shape
    dev = scores.device
    out = {
        "det_count": torch.empty((B,), dtype=torch.int32, device=dev),
        "det_box": torch.empty((B, slots, 4), dtype=torch.int32, device=dev),
        "det_score": torch.empty((B, slots), dtype=torch.float32, device=dev),
        "det_label": torch.empty((B, slots), dtype=torch.int32, device=dev),
        "crop_rect": torch.empty((B, slots, 4), dtype=torch.int32, device=dev),
        "crop_ok": torch.empty((B, slots), dtype=torch.int32, device=dev),
    }
    check(lib.yv_postprocess_dets(_p(num_dets), _p(bboxes), _p(scores), _p(labels), B, slots, _p(ratio), _p(dwdh),
                                  _p(img_wh), float(conf), float(dedupe_iou), 1 if coord_mode == "round" else 0,
                                  int(max_crops), _p(out["det_count"]), _p(out["det_box"]), _p(out["det_score"]),
                                  _p(out["det_label"]), _p(out["crop_rect"]), _p(out["crop_ok"]), _st()),
          "yv_postprocess_dets")
    return out


def compact_crops(det_count, crop_rect, crop_ok, cap: int, parts: int = 0):
    """-> crop_list (cap, 6) i32, total (1,) i32.  parts > 0: total is (1 + parts,) = {total, crops in each of `parts` equal
    slices of ceil(cap / parts) entries} (device-side: the classifier's concurrent half-batches read their counts from it)."""
    _chk_dev(det_count, crop_rect, crop_ok)
    B, slots = crop_ok.shape
    dev = crop_ok.device
    crop_list = torch.empty((max(cap, 1), 6), dtype=torch.int32, device=dev)
    total = torch.empty((1 + max(parts, 0),), dtype=torch.int32, device=dev)
    if parts > 0:
        check(lib.yv_compact_crops_split(_p(det_count), _p(crop_rect), _p(crop_ok), B, slots, cap, parts, _p(crop_list),
                                         _p(total), _st()), "yv_compact_crops_split")
    else:
        check(lib.yv_compact_crops(_p(det_count), _p(crop_rect), _p(crop_ok), B, slots, cap, _p(crop_list), _p(total),
                                   _st()), "yv_compact_crops")
    return crop_list, total


def crop_resize_norm(images: torch.Tensor, crop_list: torch.Tensor, crop_total: Optional[torch.Tensor], cap: int,
                     out_size: int = 224, patch: int = 16, layout: int = 2, out: Optional[torch.Tensor] = None):
    """images (B,H,W,3) u8 -> layout 0: (cap,3,S,S) f32 | 1: same bf16 | 2: (cap*(S/P)^2, 3*P*P) bf16."""
    _chk_dev(images, crop_list, crop_total)
    B, H, W, _ = images.shape
    dev = images.device
    if out is None:
        if layout == 0:
            out = torch.zeros((cap, 3, out_size, out_size), dtype=torch.float32, device=dev)
        elif layout == 1:
            out = torch.zeros((cap, 3, out_size, out_size), dtype=torch.bfloat16, device=dev)
        else:
            g = out_size // patch
            out = torch.zeros((cap * g * g, 3 * patch * patch), dtype=torch.bfloat16, device=dev)
    check(lib.yv_crop_resize_norm(_p(images), B, H, W, H * W * 3, _p(crop_list), _p(crop_total), cap, out_size,
                                  patch, layout, _p(out), _st()), "yv_crop_resize_norm")
    return out


def letterbox(src: torch.Tensor, geom: torch.Tensor, size: int) -> torch.Tensor:
    """src (B,Hc,Wc,3) u8 canvas, geom (B,6) i32 {w,h,nw,nh,left,top} -> (B,size,size,3) u8."""
    _chk_dev(src, geom)
    B, Hc, Wc, _ = src.shape
    out = torch.empty((B, size, size, 3), dtype=torch.uint8, device=src.device)
    check(lib.yv_letterbox(_p(src), B, Hc, Wc, _p(geom), size, _p(out), _st()), "yv_letterbox")
    return out


def augment_patchify(x: torch.Tensor, geo: torch.Tensor, idx: torch.Tensor, patch: int,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x (B,3,S,S) f32 normalised crops + one augmentation record per sample (yvhip.augment) ->
    (B*(S/P)^2, 3*P*P) bf16 patch-major rows."""
    _chk_dev(x, geo, idx)
    B, C, S, S2 = x.shape
    if C != 3 or S != S2 or x.dtype != torch.float32 or not x.is_contiguous():
        raise YvError("augment_patchify: x must be a contiguous (B,3,S,S) f32 tensor")
    if tuple(geo.shape) != (B, 6 + 2 * S) or geo.dtype != torch.float32 or not geo.is_contiguous():
        raise YvError("augment_patchify: geo must be (B, 6 + 2S) f32")
    if tuple(idx.shape) != (B, 36 + 2 * S) or idx.dtype != torch.int32 or not idx.is_contiguous():
        raise YvError("augment_patchify: idx must be (B, 36 + 2S) i32")
    g = S // patch
    if out is None:
        out = torch.empty((B * g * g, 3 * patch * patch), dtype=torch.bfloat16, device=x.device)
    check(lib.yv_augment_patchify(_p(x), B, S, patch, _p(geo), _p(idx), _p(out), _st()), "yv_augment_patchify")
    return out


def mosaic_augment(tiles: torch.Tensor, rec_f: torch.Tensor, rec_i: torch.Tensor, lut: torch.Tensor) -> torch.Tensor:
    """tiles (N,S,S,3) u8 + one record per output image (yvhip.yolo_augment) -> (B,S,S,3) u8 augmented detector inputs."""
    _chk_dev(tiles, rec_f, rec_i, lut)
    N, S, S2, C = tiles.shape
    B = rec_f.shape[0]
    if C != 3 or S != S2 or tiles.dtype != torch.uint8 or not tiles.is_contiguous():
        raise YvError("mosaic_augment: tiles must be a contiguous (N,S,S,3) u8 tensor")
    if tuple(rec_f.shape) != (B, 6) or rec_f.dtype != torch.float32 or not rec_f.is_contiguous():
        raise YvError("mosaic_augment: rec_f must be (B,6) f32")
    if tuple(rec_i.shape) != (B, 34) or rec_i.dtype != torch.int32 or not rec_i.is_contiguous():
        raise YvError("mosaic_augment: rec_i must be (B,34) i32")
    if tuple(lut.shape) != (B, 3, 256) or lut.dtype != torch.uint8 or not lut.is_contiguous():
        raise YvError("mosaic_augment: lut must be (B,3,256) u8")
    out = torch.empty((B, S, S, 3), dtype=torch.uint8, device=tiles.device)
    check(lib.yv_mosaic_augment(_p(tiles), N, B, S, _p(rec_f), _p(rec_i), _p(lut), _p(out), _st()), "yv_mosaic_augment")
    return out


def detect_decode(box_logits, cls_logits, size: int, nc: int):
    """box_logits: 3 x (B,Hs,Ws,64) f32; cls_logits: 3 x (B,Hs,Ws,ld) f32 -> boxes (B,A,4), scores (B,A,nc)."""
    _chk_dev(*box_logits, *cls_logits)
    B = box_logits[0].shape[0]
    ld = cls_logits[0].shape[-1]
    A = sum((size // s) ** 2 for s in (8, 16, 32))
    dev = box_logits[0].device
    boxes = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((B, A, nc), dtype=torch.float32, device=dev)
    check(lib.yv_detect_decode(_p(box_logits[0]), _p(box_logits[1]), _p(box_logits[2]), _p(cls_logits[0]),
                               _p(cls_logits[1]), _p(cls_logits[2]), ld, B, size, nc, _p(boxes), _p(scores), _st()),
          "yv_detect_decode")
    return boxes, scores


def detect_tail(feats, c3: int, w2, b2, w3, b3, size: int, nc: int):
    """Fused Detect tail: feats 3 x (B,Hs,Ws,ld) bf16 (box-branch features in channels 0..63, class-branch features behind them);
    w2 3 x (64,64) bf16, b2 3 x (64) f32, w3 3 x (16,c3) bf16 (rows >= nc zero), b3 3 x (16) f32 -> boxes (B,A,4), scores (B,A,nc),
    bit-identical to the two 1 x 1 convolutions + detect_decode."""
    _chk_dev(*feats, *w2, *b2, *w3, *b3)
    B, ld = feats[0].shape[0], feats[0].shape[-1]
    A = sum((size // s) ** 2 for s in (8, 16, 32))
    dev = feats[0].device
    boxes = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((B, A, nc), dtype=torch.float32, device=dev)
    arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
    check(lib.yv_detect_tail(_p(feats[0]), _p(feats[1]), _p(feats[2]), ld, c3, arr(w2), arr(b2), arr(w3), arr(b3), B, size, nc,
                             _p(boxes), _p(scores), _st()), "yv_detect_tail")
    return boxes, scores


def c2f_fused(x: torch.Tensor, c: int, n: int, w_cv1, b_cv1, w_m, b_m, w_cv2, b_cv2, out: torch.Tensor):
    """One launch for a backbone C2f block (cv1, n bottlenecks with shortcut, cv2; SiLU everywhere): x (B,H,W,2c) bf16 ->
    out (B,H,W,2c) bf16; weights (Cout, k*k*Cin) bf16 / biases f32 as conv2d takes them, w_m / b_m = [m0.cv1, m0.cv2, ...]."""
    _chk_dev(x, out, w_cv1, b_cv1, w_cv2, b_cv2, *w_m, *b_m)
    B, H, W = x.shape[0], x.shape[1], x.shape[2]
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    check(lib.yv_c2f_fused(_p(x), x.stride(2), B, H, W, c, n, _p(w_cv1), _p(b_cv1), arr(w_m), arr(b_m), _p(w_cv2), _p(b_cv2),
                           _p(out), out.stride(2), _st()), "yv_c2f_fused")
    return out


# --------------------------------------------------------------- training
def loss_fwd_bwd(logits: torch.Tensor, labels: torch.Tensor, w_lsce: float = 1.0 / 6.0, w_focal: float = 5.0 / 6.0):
    _chk_dev(logits, labels)
    B, nc = logits.shape
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
    grad = torch.empty_like(logits)
    check(lib.yv_loss_fwd_bwd(_p(logits), _p(labels), B, nc, float(w_lsce), float(w_focal), _p(loss), _p(grad), _st()),
          "yv_loss_fwd_bwd")
    return loss, grad


def sgd_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, lr: float, momentum: float = 0.9,
             weight_decay: float = 1e-3, first: bool = False, grad_scale: float = 1.0,
             mirror: Optional[torch.Tensor] = None):
    _chk_dev(p, g, m, mirror)
    check(lib.yv_sgd_step(_p(p), _p(g), _p(m), p.numel(), float(lr), float(momentum), float(weight_decay),
                          float(grad_scale), 1 if first else 0, _p(mirror), _st()), "yv_sgd_step")


def transpose_bf16_batched(src: torch.Tensor, dst: torch.Tensor, rows: int, cols: int, batch: int = 1, src_stride: int = 0,
                           dst_stride: int = 0):
    """dst[b] (cols, rows) = src[b] (rows, cols)^T for b < batch; strides in elements between consecutive matrices."""
    _chk_dev(src, dst)
    check(lib.yv_transpose_bf16_batched(_p(src), _p(dst), rows, cols, batch, src_stride, dst_stride, _st()),
          "yv_transpose_bf16_batched")


# ------------------------------------------------------------- dense math
EPI_BIAS, EPI_SILU, EPI_GELU, EPI_RES_F32, EPI_RES_BF16, EPI_OUT_F32, EPI_POSEMB = 1, 2, 4, 8, 16, 32, 64
LINEAR_HOOK = None      # callable(M, N, K, start_event, end_event) or None


_HIP = None


def _hip():
    global _HIP
    if _HIP is None:
        _HIP = C.CDLL("libamdhip64.so")
        _HIP.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        _HIP.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        _HIP.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        _HIP.hipEventDestroy.argtypes = [C.c_void_p]
        _HIP.hipEventSynchronize.argtypes = [C.c_void_p]
    return _HIP


_EVENT_POOL: list = []


def reserve_events(n: int):
    """Create n HipEvents ahead of a timed region (hipEventCreate inside it would be host time on the measured path)."""
    while len(_EVENT_POOL) < n:
        _EVENT_POOL.append(HipEvent())


def _timing_event() -> "HipEvent":
    return _EVENT_POOL.pop() if _EVENT_POOL else HipEvent()


class HipEvent:
    """Plain hipEvent_t (timing enabled) for launch-attached timestamps; elapsed_time() in ms like torch.cuda.Event."""

    def __init__(self):
        self.handle = C.c_void_p()
        if _hip().hipEventCreate(C.byref(self.handle)) != 0:
            raise YvError("hipEventCreate failed")

    def record(self, stream: Optional[int] = None):
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        if _hip().hipEventRecord(self.handle, C.c_void_p(st)) != 0:
            raise YvError("hipEventRecord failed")

    def elapsed_time(self, other: "HipEvent") -> float:
        ms = C.c_float()
        rc = _hip().hipEventElapsedTime(C.byref(ms), self.handle, other.handle)
        if rc != 0:
            raise YvError(f"hipEventElapsedTime failed ({rc})")
        return float(ms.value)

    def __del__(self):
        try:
            if self.handle:
                _hip().hipEventDestroy(self.handle)
        except Exception:
            pass


def view(t: torch.Tensor, c_off: int, c: int, up: int = 0) -> "yv_view":
    """NHWC bf16 tensor (B,H,W,ld) -> operand view of channels [c_off, c_off+c)."""
    assert t.dtype == torch.bfloat16 and t.is_cuda and t.is_contiguous()
    return yv_view(C.c_void_p(t.data_ptr() + 2 * c_off), t.shape[-1], c, up)


def linear(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, flags: int = 0,
           pos: Optional[torch.Tensor] = None, tok: int = 0, m_dev: Optional[torch.Tensor] = None, m_mul: int = 1,
           M: Optional[int] = None):
    """out[M,N] (+)= a[M,K] @ w[N,K]^T with the fused epilogue selected by `flags`."""
    _chk_dev(a, w, bias, out, pos, m_dev)
    Mr = a.shape[0] if M is None else M
    K = a.shape[1]
    N = w.shape[0]
    assert w.shape[1] == K and a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    if bias is not None:
        flags |= EPI_BIAS
    hook = LINEAR_HOOK
    if hook is not None:                     # bench.py: HIP events attached to the launch itself (hipExtLaunchKernel)
        e0, e1 = _timing_event(), _timing_event()
        check(lib.yv_set_launch_timing(e0.handle, e1.handle), "yv_set_launch_timing")
    check(lib.yv_linear(_p(a), a.stride(0), _p(w), _p(bias), Mr, N, K, _p(out), out.stride(0), _p(pos), tok, flags,
                        _p(m_dev), m_mul, _st()), "yv_linear")
    if hook is not None:
        lib.yv_set_launch_timing(None, None)             # not consumed when the call took a non-DMA kernel path
        hook(Mr, N, K, e0, e1)
    return out


_CONV_WS = {}


def _conv_workspace(device) -> torch.Tensor:
    """Split-K partial-sum workspace (64 MB f32 per device, allocated once; launches on one stream serialise on it)."""
    key = str(device)
    if key not in _CONV_WS:
        _CONV_WS[key] = torch.empty((16 * 1024 * 1024,), dtype=torch.float32, device=device)
    return _CONV_WS[key]


def conv2d(in0: "yv_view", in1: Optional["yv_view"], B: int, Hout: int, Wout: int, ksize: int, stride: int,
           weight: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, out_c_off: int, flags: int,
           res: Optional[torch.Tensor] = None, res_c_off: int = 0):
    """Conv2d + bias (+SiLU ...) ; `out` is a (B,Hout,Wout,ld) tensor, written at channel offset out_c_off."""
    Cout = weight.shape[0]
    esz = 4 if (flags & EPI_OUT_F32) else 2
    optr = C.c_void_p(out.data_ptr() + esz * out_c_off)
    rptr = None if res is None else C.c_void_p(res.data_ptr() + 2 * res_c_off)
    ws = _conv_workspace(out.device)
    check(lib.yv_conv2d_ws(C.byref(in0), C.byref(in1) if in1 is not None else None, B, Hout, Wout, ksize, stride,
                           _p(weight), _p(bias), Cout, optr, out.shape[-1], rptr,
                           0 if res is None else res.shape[-1], flags | EPI_BIAS, _p(ws), ws.numel() * 4, _st()),
          "yv_conv2d_ws")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, y: torch.Tensor, rows: int, D: int,
              ldx: int, ldy: int, eps: float = 1e-6, count_dev: Optional[torch.Tensor] = None,
              rows_per_count: int = 1):
    check(lib.yv_layernorm(_p(x), ldx, _p(gamma), _p(beta), rows, D, float(eps), _p(y), ldy, _p(count_dev),
                           rows_per_count, _st()), "yv_layernorm")
    return y


def attention(qkv: torch.Tensor, R: int, N: int, H: int, out: torch.Tensor, scale: Optional[float] = None,
              r_dev: Optional[torch.Tensor] = None):
    _chk_dev(qkv, out, r_dev)
    check(lib.yv_attention(_p(qkv), R, N, H, float(64 ** -0.5 if scale is None else scale), _p(out), _p(r_dev),
                           _st()), "yv_attention")
    return out


def cls_rows(cls: torch.Tensor, pos: torch.Tensor, R: int, tok: int, D: int, x: torch.Tensor):
    check(lib.yv_cls_rows(_p(cls), _p(pos), R, tok, D, _p(x), _st()), "yv_cls_rows")


def wrapper_head(feats: torch.Tensor, w1, b1, w2, b2, R: int, nc: int, logits: torch.Tensor, labels: torch.Tensor,
                 scale: float = 1.0, accumulate: bool = False, r_dev: Optional[torch.Tensor] = None):
    check(lib.yv_wrapper_head(_p(feats), feats.stride(0), _p(w1), _p(b1), _p(w2), _p(b2), R, nc, float(scale),
                              1 if accumulate else 0, _p(logits), _p(labels), _p(r_dev), _st()), "yv_wrapper_head")


def sppf_pool(buf: torch.Tensor, c: int):
    B, H, W, ld = buf.shape
    check(lib.yv_sppf_pool(_p(buf), B, H, W, ld, c, _st()), "yv_sppf_pool")


def stem_conv(images: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, out: torch.Tensor):
    B, H, W, _ = images.shape
    check(lib.yv_stem_conv(_p(images), B, H, W, _p(weight), _p(bias), weight.shape[1], _p(out), out.shape[-1],
                           _st()), "yv_stem_conv")


# ------------------------------------------------------------ training ops
EPI_SAVE_PRE, EPI_GELU_BWD = 128, 256


def linear_ex(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, flags: int = 0,
              res_f32: Optional[torch.Tensor] = None, aux: Optional[torch.Tensor] = None, M: Optional[int] = None):
    """Training form of `linear` (separate f32 residual source, saved pre-activation, GELU backward)."""
    _chk_dev(a, w, bias, out, res_f32, aux)
    Mr = a.shape[0] if M is None else M
    K, N = a.shape[1], w.shape[0]
    if bias is not None:
        flags |= EPI_BIAS
    check(lib.yv_linear_ex(_p(a), a.stride(0), _p(w), _p(bias), Mr, N, K, _p(out), out.stride(0), flags, _p(res_f32),
                           _p(aux), 0 if aux is None else aux.stride(0), _st()), "yv_linear_ex")
    return out


def attention_train(qkv, R, N, H, out, lse, scale=None):
    check(lib.yv_attention_train(_p(qkv), R, N, H, float(64 ** -0.5 if scale is None else scale), _p(out), _p(lse),
                                 _st()), "yv_attention_train")


def attention_bwd(qkv, out, dout, lse, R, N, H, dqkv, delta_ws, scale=None):
    check(lib.yv_attention_bwd(_p(qkv), _p(out), _p(dout), _p(lse), R, N, H, float(64 ** -0.5 if scale is None else scale),
                               _p(dqkv), _p(delta_ws), _st()), "yv_attention_bwd")


def transpose_bf16(x: torch.Tensor, out_t: torch.Tensor, rows: Optional[int] = None):
    """x (rows, cols) bf16 -> out_t (cols, ld >= round64(rows)) with zero padding."""
    r = x.shape[0] if rows is None else rows
    check(lib.yv_transpose_bf16(_p(x), r, x.shape[1], x.stride(0), _p(out_t), out_t.stride(0), _st()), "yv_transpose_bf16")
    return out_t


def cast_weights(w: torch.Tensor, w_bf16: torch.Tensor, wt_bf16: torch.Tensor):
    N, K = w.shape
    check(lib.yv_cast_weights(_p(w), N, K, _p(w_bf16), _p(wt_bf16), wt_bf16.stride(0), _st()), "yv_cast_weights")


def colsum_ws_floats(rows: int, cols: int) -> int:
    return int(lib.yv_colsum_ws_floats(rows, cols))


def cast_colsum(x: torch.Tensor, y_bf16: Optional[torch.Tensor], colsum: Optional[torch.Tensor], ws: torch.Tensor,
                accumulate: bool = False):
    rows, cols = x.shape
    check(lib.yv_cast_colsum(_p(x), rows, cols, _p(y_bf16), _p(colsum), 1 if accumulate else 0, _p(ws), _st()),
          "yv_cast_colsum")


def colsum_bf16(x: torch.Tensor, colsum: torch.Tensor, ws: torch.Tensor, rows: Optional[int] = None,
                accumulate: bool = False):
    r = x.shape[0] if rows is None else rows
    check(lib.yv_colsum_bf16(_p(x), r, x.shape[1], x.stride(0), _p(colsum), 1 if accumulate else 0, _p(ws), _st()),
          "yv_colsum_bf16")


def layernorm_bwd(x, ldx, gamma, dy, lddy, rows, D, dx, lddx, dgamma, dbeta, ws, eps=1e-6):
    check(lib.yv_layernorm_bwd(_p(x), ldx, _p(gamma), _p(dy), lddy, rows, D, float(eps), _p(dx), lddx, _p(dgamma),
                               _p(dbeta), _p(ws), _st()), "yv_layernorm_bwd")


def token_reduce(dx, R, N, D, out):
    check(lib.yv_token_reduce(_p(dx), R, N, D, _p(out), _st()), "yv_token_reduce")


def head_bwd(feats, w1t, b1, w2, dlogits, R, nc, dw1, db1, dw2, db2, dfeats, ws):
    check(lib.yv_head_bwd(_p(feats), feats.stride(0), _p(w1t), _p(b1), _p(w2), _p(dlogits), R, nc, _p(dw1), _p(db1),
                          _p(dw2), _p(db2), _p(dfeats), dfeats.stride(0), _p(ws), _st()), "yv_head_bwd")


def wgrad(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, T: Optional[int] = None):
    """dw (N,K) f32 = dy[:T]^T @ x[:T]; dy (>=T,N), x (>=T,K) bf16, T a multiple of 64 with zero tail rows."""
    for t_ in (dy, x, dw):
        if not t_.is_cuda or t_.stride(-1) != 1:
            raise YvError("wgrad operands must be device tensors with unit column stride")
    t = dy.shape[0] if T is None else T
    check(lib.yv_wgrad(_p(dy), dy.stride(0), _p(x), x.stride(0), t, dy.shape[1], x.shape[1], _p(dw), dw.stride(0), _st()),
          "yv_wgrad")
    return dw


def wgrad_conv3(dyp: torch.Tensor, xp: torch.Tensor, dw: torch.Tensor, T: int, pitch: int):
    """dw (N, 9*Cin) f32 = 3x3 / stride 1 weight gradient from operands over the zero-padded pixel grid (view_op VIEW_PAD):
    dyp (>=T, N) bf16 with zero ring / tail, xp (T, Cin) bf16 DENSE view inside a buffer that has pitch + 1 rows of finite
    values on both sides (see yv_wgrad_conv3)."""
    for t_ in (dyp, xp, dw):
        if not t_.is_cuda or t_.stride(-1) != 1:
            raise YvError("wgrad_conv3 operands must be device tensors with unit column stride")
    cin = xp.shape[1]
    if xp.stride(0) != cin or dw.shape[1] != 9 * cin:
        raise YvError("wgrad_conv3: xp must be dense (row stride Cin) and dw (N, 9*Cin)")
    check(lib.yv_wgrad_conv3(_p(dyp), dyp.stride(0), _p(xp), cin, pitch, T, dyp.shape[1], _p(dw), dw.stride(0), _st()),
          "yv_wgrad_conv3")
    return dw


def linear_nn(a: torch.Tensor, w_kn: torch.Tensor, out: torch.Tensor, flags: int = 0, aux: Optional[torch.Tensor] = None,
              M: Optional[int] = None):
    """out[M,N] = a[M,K] @ w_kn[K,N] (weight read reduction-major: dgrad on the master layout)."""
    _chk_dev(a, out, aux)
    Mr = a.shape[0] if M is None else M
    K, N = w_kn.shape
    check(lib.yv_linear_nn(_p(a), a.stride(0), _p(w_kn), w_kn.stride(0), None, Mr, N, K, _p(out), out.stride(0), flags,
                           _p(aux), 0 if aux is None else aux.stride(0), _st()), "yv_linear_nn")
    return out


# ------------------------------------------------------------- detector training (row C4)
VIEW_COPY, VIEW_ADD, VIEW_UP2, VIEW_UP2_BWD, VIEW_ZERO_INSERT, VIEW_ZERO, VIEW_PAD = 0, 1, 2, 3, 4, 5, 6


def blob_nhwc8(images: torch.Tensor, out: torch.Tensor):
    """(B,H,W,3) u8 -> (B,H,W,8) bf16 /255 with zero pad channels."""
    _chk_dev(images, out)
    check(lib.yv_blob_nhwc8(_p(images), images.numel() // 3, _p(out), _st()), "yv_blob_nhwc8")


def bn_ws_floats(T: int, Cn: int) -> int:
    return int(lib.yv_bn_ws_floats(T, Cn))


def bn_stats(z: "yv_view", T: int, mean, rstd, run_mean, run_var, ws, eps: float = 1e-3, momentum: float = 0.03):
    check(lib.yv_bn_stats(z.ptr, z.ld, T, z.c, eps, momentum, _p(mean), _p(rstd), _p(run_mean), _p(run_var), _p(ws),
                          ws.numel(), _st()), "yv_bn_stats")


def bn_act_fwd(z: "yv_view", T: int, mean, rstd, gamma, beta, out: "yv_view", res: Optional["yv_view"] = None, act: int = 1):
    check(lib.yv_bn_act_fwd(z.ptr, z.ld, T, z.c, _p(mean), _p(rstd), _p(gamma), _p(beta), res.ptr if res else None,
                            res.ld if res else 0, out.ptr, out.ld, act, _st()), "yv_bn_act_fwd")


def bn_act_bwd(da: "yv_view", z: "yv_view", T: int, mean, rstd, gamma, beta, dgamma, dbeta, dz: "yv_view", ws,
               act: int = 1, batch_stats: bool = True):
    check(lib.yv_bn_act_bwd(da.ptr, da.ld, z.ptr, z.ld, T, z.c, _p(mean), _p(rstd), _p(gamma), _p(beta), act,
                            1 if batch_stats else 0, _p(dgamma), _p(dbeta), dz.ptr, dz.ld, _p(ws), ws.numel(), _st()),
          "yv_bn_act_bwd")


def view_op(mode: int, src: Optional["yv_view"], dst: "yv_view", B: int, H: int, W: int, Cn: Optional[int] = None):
    """H, W are the dims of the SMALLER grid for the up / zero-insert modes (see include/yv_hip.h)."""
    check(lib.yv_view_op(mode, src.ptr if src is not None else None, src.ld if src is not None else 0, dst.ptr, dst.ld, B, H,
                         W, Cn if Cn is not None else dst.c, _st()), "yv_view_op")


_POOL_WS = {}


def maxpool5_bwd(x: "yv_view", dout: "yv_view", din: "yv_view", B: int, H: int, W: int):
    need = B * H * W * x.c
    dev = torch.cuda.current_device()
    ws = _POOL_WS.get(dev)
    if ws is None or ws.numel() < need:
        ws = _POOL_WS[dev] = torch.empty(need, dtype=torch.uint8, device=f"cuda:{dev}")
    check(lib.yv_maxpool5_bwd(x.ptr, x.ld, dout.ptr, dout.ld, din.ptr, din.ld, B, H, W, x.c, _p(ws), ws.numel(), _st()),
          "yv_maxpool5_bwd")


def im2col3(x: "yv_view", B: int, Hin: int, Win: int, stride: int, col: torch.Tensor):
    check(lib.yv_im2col3(x.ptr, x.ld, B, Hin, Win, x.c, stride, _p(col), _st()), "yv_im2col3")


def conv_weight_dgrad(w: torch.Tensor, Cout: int, taps: int, Cin: int, wd: torch.Tensor):
    _chk_dev(w, wd)
    check(lib.yv_conv_weight_dgrad(_p(w), Cout, taps, Cin, _p(wd), _st()), "yv_conv_weight_dgrad")


def conv_view(in0: "yv_view", B: int, Hout: int, Wout: int, ksize: int, stride: int, weight: torch.Tensor, Cout: int,
              out: "yv_view", flags: int = 0, res: Optional["yv_view"] = None, bias: Optional[torch.Tensor] = None,
              out_f32: bool = False):
    """yv_conv2d on views: out (B,Hout,Wout) rows of `out.ld` elements (bf16, or f32 with out_f32); optional bias and
    bf16 residual view (EPI_RES_BF16 accumulates gradients into a slice)."""
    if bias is not None:
        flags |= EPI_BIAS
    if out_f32:
        flags |= EPI_OUT_F32
    if res is not None:
        flags |= EPI_RES_BF16
    ws = _conv_workspace(weight.device)
    check(lib.yv_conv2d_ws(C.byref(in0), None, B, Hout, Wout, ksize, stride, _p(weight), _p(bias), Cout, out.ptr, out.ld,
                           res.ptr if res is not None else None, res.ld if res is not None else 0, flags, _p(ws),
                           ws.numel() * 4, _st()), "yv_conv2d_ws")


def mview(t: torch.Tensor, c_off: int = 0, c: Optional[int] = None) -> "yv_view":
    """(rows, ld) or (B,H,W,ld) bf16/f32 tensor -> view of channels [c_off, c_off + c)."""
    assert t.is_cuda and t.is_contiguous()
    c = t.shape[-1] - c_off if c is None else c
    return yv_view(C.c_void_p(t.data_ptr() + t.element_size() * c_off), t.shape[-1], c, 0)


def detect_loss_ws_bytes(B: int, A: int, G: int) -> int:
    return int(lib.yv_detect_loss_ws_bytes(B, A, G))


def detect_loss(box, cls, dbox, dcls, B: int, size: int, nc: int, ncp: int, gt_boxes: torch.Tensor, gt_labels: torch.Tensor,
                gt_counts: torch.Tensor, loss: torch.Tensor, ws: torch.Tensor, gains=(7.5, 0.5, 1.5)):
    """v8 detection loss + gradient; box/cls/dbox/dcls: lists of the three scales' (rows, 64) / (rows, ncp) f32 tensors."""
    _chk_dev(*box, *cls, *dbox, *dcls, gt_boxes, gt_labels, gt_counts, loss, ws)
    arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
    G = gt_boxes.shape[1]
    assert gt_boxes.dtype == torch.float32 and gt_labels.dtype == torch.int32 and gt_counts.dtype == torch.int32
    check(lib.yv_detect_loss(arr(box), arr(cls), arr(dbox), arr(dcls), B, size, nc, ncp, _p(gt_boxes), _p(gt_labels),
                             _p(gt_counts), G, gains[0], gains[1], gains[2], _p(loss), _p(ws), ws.numel() * ws.element_size(),
                             _st()), "yv_detect_loss")


OPT_SGD_NESTEROV, OPT_ADAMW = 1, 2


def optim_step(kind: int, p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: Optional[torch.Tensor], lr: float, step: int,
               beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0, grad_scale: float = 1.0,
               mirror: Optional[torch.Tensor] = None):
    """torch.optim.SGD(nesterov=True) (kind 1) / torch.optim.AdamW (kind 2) on flat fp32 buffers; step counts from 1."""
    _chk_dev(p, g, m, v, mirror)
    check(lib.yv_optim_step(kind, _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                            float(weight_decay), float(grad_scale), int(step), _p(mirror), _st()), "yv_optim_step")


def ema_update(ema: torch.Tensor, src: torch.Tensor, decay: float):
    _chk_dev(ema, src)
    check(lib.yv_ema_update(_p(ema), _p(src), ema.numel(), float(decay), _st()), "yv_ema_update")


def axpby(dst: torch.Tensor, src: torch.Tensor, a: float, b: float):
    _chk_dev(dst, src)
    check(lib.yv_axpby(_p(dst), _p(src), dst.numel(), float(a), float(b), _st()), "yv_axpby")


# ------------------------------------------------------------- MXFP8 linears (BASELINE configs[4])
def quant_mxfp8(x: torch.Tensor, q: Optional[torch.Tensor] = None, scales: Optional[torch.Tensor] = None):
    """x (rows, K) bf16 -> (q (rows, K) uint8 e4m3 bytes, scales (K/128, rows_pad, 4) uint8 E8M0, rows_pad = rows up to 256)."""
    _chk_dev(x, q, scales)
    rows, K = x.shape
    rp = (rows + 255) // 256 * 256
    q = torch.empty((rows, K), dtype=torch.uint8, device=x.device) if q is None else q
    scales = torch.zeros((K // 128, rp, 4), dtype=torch.uint8, device=x.device) if scales is None else scales
    check(lib.yv_quant_mxfp8(_p(x), x.stride(0), rows, K, _p(q), q.stride(0), _p(scales), scales.shape[1], _st()),
          "yv_quant_mxfp8")
    return q, scales


def linear_mxfp8(aq: torch.Tensor, a_scale: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor,
                 bias: Optional[torch.Tensor], out: torch.Tensor, flags: int = 0, m_dev: Optional[torch.Tensor] = None,
                 m_mul: int = 1):
    _chk_dev(aq, a_scale, wq, w_scale, bias, out, m_dev)
    M, K = aq.shape
    N = wq.shape[0]
    if bias is not None:
        flags |= EPI_BIAS
    hook = LINEAR_HOOK
    if hook is not None:
        e0, e1 = _timing_event(), _timing_event()
        check(lib.yv_set_launch_timing(e0.handle, e1.handle), "yv_set_launch_timing")
    check(lib.yv_linear_mxfp8(_p(aq), aq.stride(0), _p(a_scale), a_scale.shape[1], _p(wq), _p(w_scale), w_scale.shape[1],
                              _p(bias), M, N, K, _p(out), out.stride(0), flags, _p(m_dev), m_mul, _st()), "yv_linear_mxfp8")
    if hook is not None:
        lib.yv_set_launch_timing(None, None)
        hook(M, N, K, e0, e1)
    return out


def layernorm_mxfp8(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, q: torch.Tensor, scales: torch.Tensor, rows: int,
                    D: int, ldx: int, eps: float = 1e-6, count_dev: Optional[torch.Tensor] = None, rows_per_count: int = 1):
    """LayerNorm -> MXFP8 operand (q (rows, D) e4m3 bytes, scales (D/128, rows_pad, 4) E8M0) in one pass."""
    check(lib.yv_layernorm_mxfp8(_p(x), ldx, _p(gamma), _p(beta), rows, D, float(eps), _p(q), q.stride(0), _p(scales),
                                 scales.shape[1], _p(count_dev), rows_per_count, _st()), "yv_layernorm_mxfp8")


def linear_mxfp8_q(aq: torch.Tensor, a_scale: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor,
                   bias: Optional[torch.Tensor], out_q: torch.Tensor, out_scale: torch.Tensor, flags: int = 0,
                   m_dev: Optional[torch.Tensor] = None, m_mul: int = 1):
    """MXFP8 linear whose output is again an MXFP8 operand (bias / GELU -> bf16 rounding -> e4m3 + E8M0)."""
    _chk_dev(aq, a_scale, wq, w_scale, bias, out_q, out_scale, m_dev)
    M, K = aq.shape
    N = wq.shape[0]
    if bias is not None:
        flags |= EPI_BIAS
    hook = LINEAR_HOOK
    if hook is not None:
        e0, e1 = _timing_event(), _timing_event()
        check(lib.yv_set_launch_timing(e0.handle, e1.handle), "yv_set_launch_timing")
    check(lib.yv_linear_mxfp8_q(_p(aq), aq.stride(0), _p(a_scale), a_scale.shape[1], _p(wq), _p(w_scale), w_scale.shape[1],
                                _p(bias), M, N, K, flags, _p(m_dev), m_mul, _p(out_q), out_q.stride(0), _p(out_scale),
                                out_scale.shape[1], _st()), "yv_linear_mxfp8_q")
    if hook is not None:
        lib.yv_set_launch_timing(None, None)
        hook(M, N, K, e0, e1)


def attention_mxfp8(qkv: torch.Tensor, R: int, N: int, H: int, out_q: torch.Tensor, out_scale: torch.Tensor,
                    scale: Optional[float] = None, r_dev: Optional[torch.Tensor] = None):
    _chk_dev(qkv, out_q, out_scale, r_dev)
    check(lib.yv_attention_mxfp8(_p(qkv), R, N, H, float(64 ** -0.5 if scale is None else scale), _p(out_q), out_q.stride(0),
                                 _p(out_scale), out_scale.shape[1], _p(r_dev), _st()), "yv_attention_mxfp8")


# development knob: YV_OPTIONS="key=value,key=value" applies yv_set_option pairs at import (A/B runs of bench.py / tools without
# editing them); unknown keys fail loudly
for _kv in filter(None, os.environ.get("YV_OPTIONS", "").split(",")):
    _k, _v = _kv.split("=")
    set_option(_k.strip(), int(_v))
