"""Host-side execution plans for the two networks of the hot path.

`YoloEngine`  - YOLOv8 (n/s/m) backbone + neck + Detect head + DFL decode
`VitEngine`   - timm-layout ViT (patch 16) + Network_Wrapper head (rows B3, B4)

Both consume state dicts in the key layout of the packages the reference uses
(`ultralytics` fused model: `model.{i}...conv.weight/bias`; `Network_Wrapper(timm ViT)`:
`model.*` + `fc.1.*`/`fc.3.*`, utils/utils.py:59-87) and replay a fixed list of C-ABI
launches on the current stream.  No tensor math happens in PyTorch: torch only owns the
device buffers.  Activations are NHWC bf16; concat / split / upsample never materialise
(operand views + dual-source gather in the conv kernel).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch

from .guard import StepGuard
from . import (set_option, EPI_GELU, EPI_OUT_F32, EPI_POSEMB, EPI_RES_BF16, EPI_RES_F32, EPI_SILU, YvError, attention, attention_mxfp8,
               cls_rows,
               c2f_fused, conv2d, detect_decode, detect_tail, layernorm, layernorm_mxfp8, linear, linear_mxfp8, linear_mxfp8_q, quant_mxfp8, require_gpu, sppf_pool, stem_conv, view,
               wrapper_head)

# --------------------------------------------------------------------------------------- YOLOv8
YOLO_SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768)}
REG_MAX = 16
LAYER_STRIDE = {0: 2, 1: 4, 2: 4, 3: 8, 4: 8, 5: 16, 6: 16, 7: 32, 8: 32, 9: 32, 12: 16, 15: 8, 16: 16, 18: 16,
                19: 32, 21: 32}


def _c(ch: int, scale: str) -> int:
    _, w, mx = YOLO_SCALES[scale]
    return int(math.ceil(min(ch, mx) * w / 8) * 8)


def _n(rep: int, scale: str) -> int:
    return max(round(rep * YOLO_SCALES[scale][0]), 1)


def yolo_layers(scale: str):
    """(index, kind, params) of yolov8.yaml at this scale; `src` = skip connection layer."""
    c = lambda v: _c(v, scale)
    n = lambda v: _n(v, scale)
    return [
        (0, "stem", dict(cout=c(64))),
        (1, "conv", dict(cin=c(64), cout=c(128))),
        (2, "c2f", dict(cin=c(128), cout=c(128), n=n(3), add=True)),
        (3, "conv", dict(cin=c(128), cout=c(256))),
        (4, "c2f", dict(cin=c(256), cout=c(256), n=n(6), add=True)),
        (5, "conv", dict(cin=c(256), cout=c(512))),
        (6, "c2f", dict(cin=c(512), cout=c(512), n=n(6), add=True)),
        (7, "conv", dict(cin=c(512), cout=c(1024))),
        (8, "c2f", dict(cin=c(1024), cout=c(1024), n=n(3), add=True)),
        (9, "sppf", dict(cin=c(1024), cout=c(1024))),
        (12, "c2f", dict(cin=c(1024) + c(512), cout=c(512), n=n(3), add=False, a=(9, 1), b=(6, 0))),
        (15, "c2f", dict(cin=c(512) + c(256), cout=c(256), n=n(3), add=False, a=(12, 1), b=(4, 0))),
        (16, "conv", dict(cin=c(256), cout=c(256))),
        (18, "c2f", dict(cin=c(256) + c(512), cout=c(512), n=n(3), add=False, a=(16, 0), b=(12, 0))),
        (19, "conv", dict(cin=c(512), cout=c(512))),
        (21, "c2f", dict(cin=c(512) + c(1024), cout=c(1024), n=n(3), add=False, a=(19, 0), b=(9, 0))),
    ]


def yolo_conv_keys(scale: str, nc: int) -> List[Tuple[str, int, int, int]]:
    """(state-dict prefix, cin, cout, k) of every conv in the fused model."""
    out = []
    for idx, kind, p in yolo_layers(scale):
        pre = f"model.{idx}."
        if kind == "stem":
            out.append((pre + "conv", 3, p["cout"], 3))
        elif kind == "conv":
            out.append((pre + "conv", p["cin"], p["cout"], 3))
        elif kind == "c2f":
            c = p["cout"] // 2
            out.append((pre + "cv1.conv", p["cin"], 2 * c, 1))
            out.append((pre + "cv2.conv", (2 + p["n"]) * c, p["cout"], 1))
            for j in range(p["n"]):
                out.append((pre + f"m.{j}.cv1.conv", c, c, 3))
                out.append((pre + f"m.{j}.cv2.conv", c, c, 3))
        elif kind == "sppf":
            out.append((pre + "cv1.conv", p["cin"], p["cin"] // 2, 1))
            out.append((pre + "cv2.conv", p["cin"] * 2, p["cout"], 1))
    ch = (_c(256, scale), _c(512, scale), _c(1024, scale))
    c2 = max(16, ch[0] // 4, REG_MAX * 4)
    c3 = max(ch[0], min(nc, 100))
    for s, ci in enumerate(ch):
        out += [(f"model.22.cv2.{s}.0.conv", ci, c2, 3), (f"model.22.cv2.{s}.1.conv", c2, c2, 3),
                (f"model.22.cv2.{s}.2", c2, 4 * REG_MAX, 1), (f"model.22.cv3.{s}.0.conv", ci, c3, 3),
                (f"model.22.cv3.{s}.1.conv", c3, c3, 3), (f"model.22.cv3.{s}.2", c3, nc, 1)]
    return out


def init_yolo_state(scale: str = "n", nc: int = 5, seed: int = 42, head_gain: float = 1.0,
                    cls_bias: float = 0.0) -> Dict[str, torch.Tensor]:
    """Seeded random fused weights (BN folded to identity) in the ultralytics key layout.
    There is no network on the box: checkpoints cannot be fetched, so benchmarks and
    smoke tests use this.  `head_gain` widens the spread of the Detect outputs."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for key, ci, co, k in yolo_conv_keys(scale, nc):
        gain = head_gain if key.endswith(".2") else 1.0
        sd[key + ".weight"] = torch.randn(co, ci, k, k, generator=g) * (gain * math.sqrt(2.0 / (ci * k * k)))
        sd[key + ".bias"] = torch.randn(co, generator=g) * 0.1
        if key.startswith("model.22.cv3.") and key.endswith(".2"):
            sd[key + ".bias"] += cls_bias
    sd["model.22.dfl.conv.weight"] = torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1)
    return sd


def _khwc(w: torch.Tensor) -> torch.Tensor:
    """(Cout,Cin,k,k) f32 -> (Cout, k*k*Cin) bf16 with K order (ky,kx,cin)."""
    co = w.shape[0]
    return w.permute(0, 2, 3, 1).reshape(co, -1).contiguous().to(torch.bfloat16)


class YoloEngine:
    """images (B,S,S,3) u8 RGB on the device -> boxes (B,A,4) f32 xyxy, scores (B,A,nc) f32."""

    def __init__(self, state: Dict[str, torch.Tensor], scale: str = "n", nc: int = 5, size: int = 640,
                 device: str = "cuda:0"):
        require_gpu()
        if size % 32:
            raise YvError("input size must be a multiple of 32")
        self.scale, self.nc, self.size, self.dev = scale, nc, size, torch.device(device)
        self.layers = yolo_layers(scale)
        self.w: Dict[str, torch.Tensor] = {}
        self.b: Dict[str, torch.Tensor] = {}
        need = yolo_conv_keys(scale, nc)
        for key, ci, co, k in need:
            wt, bs = state[key + ".weight"], state[key + ".bias"]
            if tuple(wt.shape) != (co, ci, k, k):
                raise YvError(f"{key}.weight has shape {tuple(wt.shape)}, expected {(co, ci, k, k)}")
            if key == "model.0.conv":
                self.w[key] = wt.float().permute(2, 3, 1, 0).reshape(27, co).contiguous().to(self.dev)    # (tap*3+c, cout)
            else:
                self.w[key] = _khwc(wt.float()).to(self.dev)
            self.b[key] = bs.float().contiguous().to(self.dev)
        ch = (_c(256, scale), _c(512, scale), _c(1024, scale))
        self.c2 = max(16, ch[0] // 4, REG_MAX * 4)
        self.c3 = max(ch[0], min(nc, 100))
        self.ncp = (nc + 7) // 8 * 8
        for s in range(3):
            # horizontal fusion of the two 3x3 convs that share the scale's feature map (test.ipynb:1285)
            k0, k1 = f"model.22.cv2.{s}.0.conv", f"model.22.cv3.{s}.0.conv"
            self.w[f"det{s}.0"] = torch.cat([self.w[k0], self.w[k1]], 0).contiguous()
            self.b[f"det{s}.0"] = torch.cat([self.b[k0], self.b[k1]], 0).contiguous()
            kc = f"model.22.cv3.{s}.2"                       # class logits: pad Cout to a multiple of 8
            wp = torch.zeros(self.ncp, self.c3, dtype=torch.bfloat16, device=self.dev)
            wp[:nc] = self.w[kc]
            bp = torch.zeros(self.ncp, dtype=torch.float32, device=self.dev)
            bp[:nc] = self.b[kc]
            self.w[kc + ".pad"], self.b[kc + ".pad"] = wp, bp
            wp16 = torch.zeros(16, self.c3, dtype=torch.bfloat16, device=self.dev)     # operand of the fused tail: one MFMA row fragment
            bp16 = torch.zeros(16, dtype=torch.float32, device=self.dev)
            if nc <= 16:
                wp16[:nc] = self.w[kc]
                bp16[:nc] = self.b[kc]
            self.w[kc + ".pad16"], self.b[kc + ".pad16"] = wp16, bp16
        # fused Detect tail (yv_detect_tail: last 1 x 1 convolutions + DFL decode + sigmoid in one launch, bit-identical)
        self.fused_tail = self.c2 == 64 and nc <= 16 and self.c3 in (64, 128, 192)
        self.fused_c2f = True                    # backbone C2f blocks with c <= 32 in one launch each (yv_c2f_fused)
        self._bufs: Dict[int, dict] = {}
        self.guard = StepGuard()                 # one replay of the launch list at a time (callers may be threads)
        self.A = sum((size // s) ** 2 for s in (8, 16, 32))

    # -- buffers are allocated once per batch size and reused (no allocation in the step)
    def _buffers(self, B: int) -> dict:
        if B in self._bufs:
            return self._bufs[B]
        S = self.size
        bf = lambda h, c: torch.zeros((B, h, h, c), dtype=torch.bfloat16, device=self.dev)
        f32 = lambda h, c: torch.zeros((B, h, h, c), dtype=torch.float32, device=self.dev)
        bufs: dict = {"out": {}, "y": {}, "t": {}}
        bufs["h"] = {}
        for idx, kind, p in self.layers:
            h = S // LAYER_STRIDE[idx]
            bufs["out"][idx] = bf(h, p["cout"])
            bufs["h"][idx] = h
            if kind == "c2f":
                c = p["cout"] // 2
                bufs["y"][idx] = bf(h, (2 + p["n"]) * c)
                bufs["t"][idx] = bf(h, c)
            elif kind == "sppf":
                bufs["y"][idx] = bf(h, p["cin"] * 2)
        for s, st in enumerate((8, 16, 32)):
            hs = S // st
            bufs[f"det{s}.hb"] = bf(hs, self.c2 + self.c3)
            bufs[f"det{s}.hc"] = bf(hs, self.c2 + self.c3)
            bufs[f"det{s}.box"] = f32(hs, 4 * REG_MAX)
            bufs[f"det{s}.cls"] = f32(hs, self.ncp)
        self._bufs[B] = bufs
        return bufs

    def _c2f(self, idx: int, p: dict, in0, in1, B: int, bufs: dict):
        pre = f"model.{idx}."
        h = bufs["h"][idx]
        c = p["cout"] // 2
        y, t, out = bufs["y"][idx], bufs["t"][idx], bufs["out"][idx]
        if (self.fused_c2f and in1 is None and p["add"] and p["cin"] == p["cout"] and c in (16, 32) and p["n"] in (1, 2)):
            # the whole block in one launch, intermediates in LDS (yv_c2f_fused; YOLOv8n: model.2 and model.4)
            keys = [pre + f"m.{j}.cv{k}.conv" for j in range(p["n"]) for k in (1, 2)]
            c2f_fused(bufs["out"][idx - 1], c, p["n"], self.w[pre + "cv1.conv"], self.b[pre + "cv1.conv"],
                      [self.w[k] for k in keys], [self.b[k] for k in keys], self.w[pre + "cv2.conv"], self.b[pre + "cv2.conv"], out)
            return
        conv2d(in0, in1, B, h, h, 1, 1, self.w[pre + "cv1.conv"], self.b[pre + "cv1.conv"], y, 0, EPI_SILU)
        for j in range(p["n"]):
            src = (1 + j) * c
            conv2d(view(y, src, c), None, B, h, h, 3, 1, self.w[pre + f"m.{j}.cv1.conv"], self.b[pre + f"m.{j}.cv1.conv"],
                   t, 0, EPI_SILU)
            if p["add"]:
                conv2d(view(t, 0, c), None, B, h, h, 3, 1, self.w[pre + f"m.{j}.cv2.conv"],
                       self.b[pre + f"m.{j}.cv2.conv"], y, src + c, EPI_SILU | EPI_RES_BF16, res=y, res_c_off=src)
            else:
                conv2d(view(t, 0, c), None, B, h, h, 3, 1, self.w[pre + f"m.{j}.cv2.conv"],
                       self.b[pre + f"m.{j}.cv2.conv"], y, src + c, EPI_SILU)
        conv2d(view(y, 0, (2 + p["n"]) * c), None, B, h, h, 1, 1, self.w[pre + "cv2.conv"], self.b[pre + "cv2.conv"],
               out, 0, EPI_SILU)

    def forward_raw(self, images: torch.Tensor):
        """Runs backbone+neck+head; returns per-scale (box logits f32, class logits f32) NHWC tensors.  Called with `self.guard`
        held (as __call__ does) the engine-owned buffers themselves are returned - valid until the caller leaves the guard;
        called bare, the result is CLONED under the guard, so that another thread's next replay cannot overwrite what this
        caller is still reading."""
        outer = self.guard.held
        with self.guard:
            res = self._forward_raw(images)
            if outer:
                return res
            return tuple([t.clone() for t in part] for part in res)

    def _forward_raw(self, images: torch.Tensor, tail: bool = True):
        """tail=False: stop in front of the last 1 x 1 convolutions and return the per-scale feature buffers (B,Hs,Ws,c2+c3)."""
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3:
            raise YvError("images must be (B,S,S,3) uint8")
        B, S = images.shape[0], images.shape[1]
        if S != self.size or images.shape[2] != self.size:
            raise YvError(f"engine built for {self.size}x{self.size}")
        bufs = self._buffers(B)
        o = bufs["out"]
        for idx, kind, p in self.layers:
            pre = f"model.{idx}."
            h = bufs["h"][idx]
            if kind == "stem":
                stem_conv(images, self.w[pre + "conv"], self.b[pre + "conv"], o[0])
            elif kind == "conv":
                src = o[idx - 1]
                conv2d(view(src, 0, p["cin"]), None, B, h, h, 3, 2, self.w[pre + "conv"], self.b[pre + "conv"], o[idx],
                       0, EPI_SILU)
            elif kind == "c2f":
                if "a" in p:
                    (ia, ua), (ib, ub) = p["a"], p["b"]
                    in0 = view(o[ia], 0, o[ia].shape[-1], up=ua)
                    in1 = view(o[ib], 0, o[ib].shape[-1], up=ub)
                else:
                    in0, in1 = view(o[idx - 1], 0, p["cin"]), None
                self._c2f(idx, p, in0, in1, B, bufs)
            elif kind == "sppf":
                y = bufs["y"][idx]
                c_ = p["cin"] // 2
                conv2d(view(o[idx - 1], 0, p["cin"]), None, B, h, h, 1, 1, self.w[pre + "cv1.conv"],
                       self.b[pre + "cv1.conv"], y, 0, EPI_SILU)
                sppf_pool(y, c_)
                conv2d(view(y, 0, 4 * c_), None, B, h, h, 1, 1, self.w[pre + "cv2.conv"], self.b[pre + "cv2.conv"],
                       o[idx], 0, EPI_SILU)
        box_l, cls_l = [], []
        c2, c3 = self.c2, self.c3
        for s, fidx in enumerate((15, 18, 21)):
            f = o[fidx]
            hs = bufs["h"][fidx]
            hb, hc = bufs[f"det{s}.hb"], bufs[f"det{s}.hc"]
            conv2d(view(f, 0, f.shape[-1]), None, B, hs, hs, 3, 1, self.w[f"det{s}.0"], self.b[f"det{s}.0"], hb, 0,
                   EPI_SILU)
            k = f"model.22.cv2.{s}.1.conv"
            conv2d(view(hb, 0, c2), None, B, hs, hs, 3, 1, self.w[k], self.b[k], hc, 0, EPI_SILU)
            k = f"model.22.cv3.{s}.1.conv"
            conv2d(view(hb, c2, c3), None, B, hs, hs, 3, 1, self.w[k], self.b[k], hc, c2, EPI_SILU)
            if not tail:
                box_l.append(hc)
                continue
            k = f"model.22.cv2.{s}.2"
            conv2d(view(hc, 0, c2), None, B, hs, hs, 1, 1, self.w[k], self.b[k], bufs[f"det{s}.box"], 0, EPI_OUT_F32)
            k = f"model.22.cv3.{s}.2.pad"
            conv2d(view(hc, c2, c3), None, B, hs, hs, 1, 1, self.w[k], self.b[k], bufs[f"det{s}.cls"], 0, EPI_OUT_F32)
            box_l.append(bufs[f"det{s}.box"])
            cls_l.append(bufs[f"det{s}.cls"])
        return box_l, cls_l

    def __call__(self, images: torch.Tensor):
        with self.guard:                         # the decode reads the engine-owned head buffers
            if self.fused_tail:
                feats, _ = self._forward_raw(images, tail=False)
                return detect_tail(feats, self.c3,
                                   [self.w[f"model.22.cv2.{s}.2"] for s in range(3)], [self.b[f"model.22.cv2.{s}.2"] for s in range(3)],
                                   [self.w[f"model.22.cv3.{s}.2.pad16"] for s in range(3)],
                                   [self.b[f"model.22.cv3.{s}.2.pad16"] for s in range(3)], self.size, self.nc)
            box_l, cls_l = self._forward_raw(images)
            return detect_decode(box_l, cls_l, self.size, self.nc)


# ------------------------------------------------------------------------------------------ ViT
VIT_CFGS = {
    "vit_base_patch16_224": (16, 768, 12, 12),
    "vit_base_patch8_224": (8, 768, 12, 12),            # the reference's configured model (utils/class_config.py:21)
    "vit_large_patch16_224": (16, 1024, 24, 16),        # BASELINE.json configs[4]
    "vit_large_patch16_224": (16, 1024, 24, 16),
    "vit_tiny_test": (16, 128, 2, 2),
    "vit_tiny8_test": (8, 128, 2, 2),
}


def vit_cfg(name: str):
    base = name.split(".")[0]
    if base not in VIT_CFGS:
        raise YvError(f"classifier '{name}' is not supported by the MI355X path (supported: {sorted(VIT_CFGS)})")
    return VIT_CFGS[base]


def init_vit_wrapper_state(name: str, num_classes: int = 5, seed: int = 42, img: int = 224) -> Dict[str, torch.Tensor]:
    """Seeded random Network_Wrapper(timm ViT) state dict (`model.*` + `fc.1.*`, `fc.3.*`)."""
    P, D, L, H = vit_cfg(name)
    n = (img // P) ** 2 + 1
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, scale=0.02: torch.randn(*s, generator=g) * scale
    sd = {"model.cls_token": rn(1, 1, D), "model.pos_embed": rn(1, n, D),
          "model.patch_embed.proj.weight": rn(D, 3, P, P), "model.patch_embed.proj.bias": rn(D)}
    for i in range(L):
        p = f"model.blocks.{i}."
        sd[p + "norm1.weight"] = 1 + rn(D); sd[p + "norm1.bias"] = rn(D)
        sd[p + "attn.qkv.weight"] = rn(3 * D, D, scale=0.04); sd[p + "attn.qkv.bias"] = rn(3 * D)
        sd[p + "attn.proj.weight"] = rn(D, D); sd[p + "attn.proj.bias"] = rn(D)
        sd[p + "norm2.weight"] = 1 + rn(D); sd[p + "norm2.bias"] = rn(D)
        sd[p + "mlp.fc1.weight"] = rn(4 * D, D); sd[p + "mlp.fc1.bias"] = rn(4 * D)
        sd[p + "mlp.fc2.weight"] = rn(D, 4 * D); sd[p + "mlp.fc2.bias"] = rn(D)
    sd["model.norm.weight"] = 1 + rn(D); sd["model.norm.bias"] = rn(D)
    sd["model.head.weight"] = rn(1000, D, scale=0.05); sd["model.head.bias"] = rn(1000, scale=0.5)
    sd["fc.1.weight"] = rn(128, 1000, scale=0.05); sd["fc.1.bias"] = rn(128, scale=0.1)
    sd["fc.3.weight"] = rn(num_classes, 128, scale=0.2); sd["fc.3.bias"] = rn(num_classes, scale=0.1)
    return sd


class VitEngine:
    """Patch-major bf16 crops (cap*tok, 3*P*P) -> backbone logits (cap, 1024-padded) f32 and,
    through the Network_Wrapper head, class logits (cap, nc) + labels (cap)."""

    def __init__(self, state: Dict[str, torch.Tensor], name: str, num_classes: int = 5, img: int = 224,
                 device: str = "cuda:0", dtype: str = "bf16"):
        """dtype "bf16" (default) or "mxfp8": the four block linears (qkv, proj, fc1, fc2) then run on OCP e4m3 operands
        with one E8M0 scale per 32 K elements through the block-scaled MFMA (BASELINE.json configs[4]); weights are
        quantised once here, activations by yv_quant_mxfp8 in front of each GEMM; everything else (patch-embed, LayerNorm,
        attention, residual stream, heads) keeps its bf16 / f32 form."""
        require_gpu()
        if dtype not in ("bf16", "mxfp8"):
            raise YvError("dtype must be 'bf16' or 'mxfp8'")
        self.dtype = dtype
        self.fuse_attention_quant = os.environ.get("YV_MX_ATTN_FUSED", "1") == "1"     # A/B switch of the mxfp8 path
        self.P, self.D, self.L, self.H = vit_cfg(name)
        if self.D // self.H != 64:
            raise YvError("attention kernel is specialised for head dim 64")
        self.name, self.nc, self.img, self.dev = name, num_classes, img, torch.device(device)
        self.tok = (img // self.P) ** 2
        self.N = self.tok + 1
        dev, D = self.dev, self.D
        bf = lambda t: t.float().contiguous().to(torch.bfloat16).to(dev)
        f32 = lambda t: t.float().contiguous().to(dev)
        g = lambda k: state["model." + k]
        self.w_pe = bf(g("patch_embed.proj.weight").reshape(D, -1))
        self.b_pe = f32(g("patch_embed.proj.bias"))
        self.cls = f32(g("cls_token").reshape(D))
        self.pos = f32(g("pos_embed").reshape(self.N, D))
        self.blocks = []
        for i in range(self.L):
            p = f"blocks.{i}."
            self.blocks.append(dict(
                n1w=f32(g(p + "norm1.weight")), n1b=f32(g(p + "norm1.bias")),
                wqkv=bf(g(p + "attn.qkv.weight")), bqkv=f32(g(p + "attn.qkv.bias")),
                wproj=bf(g(p + "attn.proj.weight")), bproj=f32(g(p + "attn.proj.bias")),
                n2w=f32(g(p + "norm2.weight")), n2b=f32(g(p + "norm2.bias")),
                wfc1=bf(g(p + "mlp.fc1.weight")), bfc1=f32(g(p + "mlp.fc1.bias")),
                wfc2=bf(g(p + "mlp.fc2.weight")), bfc2=f32(g(p + "mlp.fc2.bias"))))
        if dtype == "mxfp8":
            if D % 128:
                raise YvError("mxfp8 needs an embedding width that is a multiple of 128")
            for blk in self.blocks:
                for k in ("wqkv", "wproj", "wfc1", "wfc2"):
                    blk[k + "_q"], blk[k + "_s"] = quant_mxfp8(blk[k])
                    del blk[k]
        self.nw, self.nb = f32(g("norm.weight")), f32(g("norm.bias"))
        wh = torch.zeros(1024, D)
        wh[:1000] = g("head.weight").float()
        bh = torch.zeros(1024)
        bh[:1000] = g("head.bias").float()
        self.w_head, self.b_head = bf(wh), f32(bh)
        self.fc1w, self.fc1b = f32(state["fc.1.weight"].float().t()), f32(state["fc.1.bias"])     # (1000,128): transposed
        self.fc2w, self.fc2b = f32(state["fc.3.weight"]), f32(state["fc.3.bias"])
        if tuple(self.fc2w.shape) != (num_classes, 128):
            raise YvError("fc.3.weight does not match num_classes")
        self._bufs: Dict[tuple, dict] = {}
        self._guards: Dict[int, StepGuard] = {}
        # PipelinedRunner: first block whose persistent GEMMs take every CU again (the reduced budget exists for the detector of the
        # NEXT batch, which runs beside the first part of a classifier pass only); None = one budget for the whole pass
        self.full_cus_from: Optional[int] = None

    def guard(self, slot: int = 0) -> StepGuard:
        """Guard of one buffer set: hold it across backbone() + head() (the features live in engine-owned buffers).
        Different slots are independent (the split classifier runs two of them concurrently on two streams)."""
        g = self._guards.get(slot)
        if g is None:
            g = self._guards.setdefault(slot, StepGuard())
        return g

    def _buffers(self, cap: int, slot: int = 0) -> dict:
        """Activation buffers for `cap` crops; `slot` selects an independent set (concurrent sub-batches)."""
        key = (cap, slot)
        if key not in self._bufs:
            dev, D, N = self.dev, self.D, self.N
            z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
            self._bufs[key] = dict(
                x=z((cap * N, D), torch.float32), h=z((cap * N, D), torch.bfloat16),
                qkv=z((cap * N, 3 * D), torch.bfloat16), o=z((cap * N, D), torch.bfloat16),
                g=z((cap * N, 4 * D), torch.bfloat16), c=z((cap, D), torch.bfloat16),
                **({} if self.dtype != "mxfp8" else dict(          # MXFP8 operand images: bytes + K-step-major block scales
                    q=z((cap * N, D), torch.uint8), qs=z((D // 128, (cap * N + 255) // 256 * 256, 4), torch.uint8),
                    gq=z((cap * N, 4 * D), torch.uint8),
                    gs=z((4 * D // 128, (cap * N + 255) // 256 * 256, 4), torch.uint8))),
                feats=z((cap, 1024), torch.float32))
        return self._bufs[key]

    def patch_buffer(self, cap: int, slot: int = 0) -> torch.Tensor:
        b = self._buffers(cap, slot)
        if "pm" not in b:
            b["pm"] = torch.zeros((cap * self.tok, 3 * self.P * self.P), dtype=torch.bfloat16, device=self.dev)
        return b["pm"]

    def backbone(self, patches: torch.Tensor, cap: int, count: Optional[torch.Tensor] = None, slot: int = 0) -> torch.Tensor:
        """patches (cap*tok, 3*P*P) bf16 -> feats (cap,1024) f32 (columns >= 1000 are zero padding)."""
        outer = self.guard(slot).held
        with self.guard(slot):
            feats = self._backbone(patches, cap, count, slot)
            return feats if outer else feats.clone()         # bare call: a copy that the next replay cannot overwrite

    def _backbone(self, patches: torch.Tensor, cap: int, count: Optional[torch.Tensor], slot: int) -> torch.Tensor:
        b = self._buffers(cap, slot)
        D, N, tok, H = self.D, self.N, self.tok, self.H
        x, h, qkv, o, gbuf = b["x"], b["h"], b["qkv"], b["o"], b["g"]
        cls_rows(self.cls, self.pos, cap, tok, D, x)
        linear(patches, self.w_pe, self.b_pe, x, flags=EPI_OUT_F32 | EPI_POSEMB, pos=self.pos, tok=tok, m_dev=count,
               m_mul=tok)
        rows = cap * N
        if self.dtype == "mxfp8":
            return self._backbone_mxfp8(b, cap, count)
        for i, blk in enumerate(self.blocks):
            if self.full_cus_from is not None and i == self.full_cus_from:
                set_option("linear_p8_cus", 0)          # the caller (PipelinedRunner) restores its own setting after the pass
            layernorm(x, blk["n1w"], blk["n1b"], h, rows, D, D, D, count_dev=count, rows_per_count=N)
            linear(h, blk["wqkv"], blk["bqkv"], qkv, m_dev=count, m_mul=N)
            attention(qkv, cap, N, H, o, r_dev=count)
            linear(o, blk["wproj"], blk["bproj"], x, flags=EPI_RES_F32, m_dev=count, m_mul=N)
            layernorm(x, blk["n2w"], blk["n2b"], h, rows, D, D, D, count_dev=count, rows_per_count=N)
            linear(h, blk["wfc1"], blk["bfc1"], gbuf, flags=EPI_GELU, m_dev=count, m_mul=N)
            linear(gbuf, blk["wfc2"], blk["bfc2"], x, flags=EPI_RES_F32, m_dev=count, m_mul=N)
        layernorm(x, self.nw, self.nb, b["c"], cap, D, N * D, D, count_dev=count, rows_per_count=1)
        linear(b["c"], self.w_head, self.b_head, b["feats"], flags=EPI_OUT_F32, m_dev=count, m_mul=1)
        return b["feats"]

    def _backbone_mxfp8(self, b: dict, cap: int, count: Optional[torch.Tensor]) -> torch.Tensor:
        """Block linears in MXFP8.  Operand hand-offs: LayerNorm writes the qkv / fc1 operand directly, the fc1 epilogue
        writes the fc2 operand directly (GELU output never exists in bf16 in HBM), attention writes the proj operand
        directly: no separate quantisation pass is left."""
        D, N, H = self.D, self.N, self.H
        x, qkv, o = b["x"], b["qkv"], b["o"]
        hq, hs, gq, gs = b["q"], b["qs"], b["gq"], b["gs"]
        rows = cap * N
        for blk in self.blocks:
            layernorm_mxfp8(x, blk["n1w"], blk["n1b"], hq, hs, rows, D, D, count_dev=count, rows_per_count=N)
            linear_mxfp8(hq, hs, blk["wqkv_q"], blk["wqkv_s"], blk["bqkv"], qkv, m_dev=count, m_mul=N)
            if H % 2 == 0 and self.fuse_attention_quant:
                attention_mxfp8(qkv, cap, N, H, hq, hs, r_dev=count)          # attention writes the proj operand directly
            else:
                attention(qkv, cap, N, H, o, r_dev=count)
                quant_mxfp8(o, hq, hs)
            linear_mxfp8(hq, hs, blk["wproj_q"], blk["wproj_s"], blk["bproj"], x, flags=EPI_RES_F32, m_dev=count, m_mul=N)
            layernorm_mxfp8(x, blk["n2w"], blk["n2b"], hq, hs, rows, D, D, count_dev=count, rows_per_count=N)
            linear_mxfp8_q(hq, hs, blk["wfc1_q"], blk["wfc1_s"], blk["bfc1"], gq, gs, flags=EPI_GELU, m_dev=count, m_mul=N)
            linear_mxfp8(gq, gs, blk["wfc2_q"], blk["wfc2_s"], blk["bfc2"], x, flags=EPI_RES_F32, m_dev=count, m_mul=N)
        layernorm(x, self.nw, self.nb, b["c"], cap, D, N * D, D, count_dev=count, rows_per_count=1)
        linear(b["c"], self.w_head, self.b_head, b["feats"], flags=EPI_OUT_F32, m_dev=count, m_mul=1)
        return b["feats"]

    def head(self, feats: torch.Tensor, cap: int, logits: torch.Tensor, labels: torch.Tensor, scale: float = 1.0,
             accumulate: bool = False, count: Optional[torch.Tensor] = None):
        wrapper_head(feats, self.fc1w, self.fc1b, self.fc2w, self.fc2b, cap, self.nc, logits, labels, scale, accumulate,
                     r_dev=count)
