"""nn.Module shells that give the HIP engines the reference's object model.

The reference builds `Network_Wrapper(timm.create_model(name, pretrained=False, num_classes=1000), nc)`
and then `.load_state_dict(torch.load(path))`, `.to(device)`, `.eval()`, `net(x)` (utils/utils.py:59-87,
app.py:35-37).  `ViTBackbone` reproduces the PARAMETER TREE of a timm VisionTransformer (same state_dict
keys and shapes) without any of its math; the forward of the wrapper goes through `VitEngine`
(hand-written kernels).  The parameters stay ordinary fp32 `nn.Parameter`s so checkpoints round-trip
through `state_dict()` / `load_state_dict()` byte for byte; the bf16 device copies the kernels read are
rebuilt lazily when the parameters change (version counters).
"""
from __future__ import annotations

import threading
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import YvError, require_gpu
from .engines import VitEngine, vit_cfg


_ENGINE_BUILD_LOCK = threading.Lock()      # module-level: a lock attribute would break copy / pickling of the nn.Module


class _Attn(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.qkv = nn.Linear(D, 3 * D)
        self.proj = nn.Linear(D, D)


class _Mlp(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.fc1 = nn.Linear(D, 4 * D)
        self.fc2 = nn.Linear(4 * D, D)


class _Block(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.norm1 = nn.LayerNorm(D, eps=1e-6)
        self.attn = _Attn(D)
        self.norm2 = nn.LayerNorm(D, eps=1e-6)
        self.mlp = _Mlp(D)


class _PatchEmbed(nn.Module):
    def __init__(self, P, D):
        super().__init__()
        self.proj = nn.Conv2d(3, D, kernel_size=P, stride=P)


class ViTBackbone(nn.Module):
    """Parameter tree of timm's VisionTransformer (cls_token, pos_embed, patch_embed.proj, blocks.*, norm, head)."""

    def __init__(self, name: str, num_classes: int = 1000, img: int = 224):
        super().__init__()
        P, D, L, H = vit_cfg(name)
        self.arch = name
        self.img = img
        n = (img // P) ** 2 + 1
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.pos_embed = nn.Parameter(torch.randn(1, n, D) * 0.02)
        self.patch_embed = _PatchEmbed(P, D)
        self.blocks = nn.ModuleList([_Block(D) for _ in range(L)])
        self.norm = nn.LayerNorm(D, eps=1e-6)
        self.head = nn.Linear(D, num_classes)

    def forward(self, x):
        raise YvError("ViTBackbone has no standalone forward: call it through Network_Wrapper (HIP engine)")


def create_model(name: str, pretrained: bool = False, num_classes: int = 1000) -> ViTBackbone:
    """Stand-in for `timm.create_model(name, pretrained=False, num_classes=1000)` (utils/utils.py:82)."""
    if pretrained:
        raise YvError("pretrained=True would fetch weights from the network; the reference never does (utils/utils.py:82)")
    if num_classes != 1000:
        raise YvError("Network_Wrapper.fc expects a 1000-d backbone output (utils/utils.py:66)")
    return ViTBackbone(name, num_classes)


def patchify_bf16(x: torch.Tensor, P: int) -> torch.Tensor:
    """(R,3,S,S) float tensor on the device -> (R*(S/P)^2, 3*P*P) bf16 rows (data movement only)."""
    R, C, S, _ = x.shape
    g = S // P
    return x.reshape(R, C, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(R * g * g, C * P * P).to(torch.bfloat16).contiguous()


class Network_Wrapper(nn.Module):
    """Same constructor, attributes (`model`, `fc`) and state_dict keys as utils/utils.py:59-72 /
    utils/trainClass.py:26-42; `forward(x)` = fc(model(x)) evaluated by the HIP engine."""

    def __init__(self, model, num_class):
        super().__init__()
        self.model = model
        hidden_units = 128
        self.fc = nn.Sequential(nn.ReLU(), nn.Linear(1000, hidden_units), nn.ReLU(), nn.Linear(hidden_units, num_class))
        self.num_class = num_class
        self._engine: Optional[VitEngine] = None
        self._engine_key = None

    def _state_key(self):
        return tuple((id(p), p._version, str(p.device)) for p in self.parameters())

    def engine(self) -> VitEngine:
        if not isinstance(self.model, ViTBackbone):
            raise YvError("Network_Wrapper.model must come from yvhip.modules.create_model (timm is not used)")
        with _ENGINE_BUILD_LOCK:                 # concurrent first calls must not build (and then swap) two engines
            key = self._state_key()
            if self._engine is None or key != self._engine_key:
                require_gpu()
                dev = next(self.parameters()).device
                if dev.type != "cuda":
                    dev = torch.device("cuda", torch.cuda.current_device())
                sd = {k: v.detach() for k, v in self.state_dict().items()}
                self._engine = VitEngine(sd, self.model.arch, self.num_class, self.model.img, device=str(dev))
                self._engine_key = key
            return self._engine

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        eng = self.engine()
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != eng.img or x.shape[3] != eng.img:
            raise YvError(f"expected (R,3,{eng.img},{eng.img}) input, got {tuple(x.shape)}")
        R = x.shape[0]
        x = x.to(eng.dev)
        patches = patchify_bf16(x.float(), eng.P)
        logits = torch.zeros((R, self.num_class), dtype=torch.float32, device=eng.dev)
        labels = torch.zeros((R,), dtype=torch.int32, device=eng.dev)
        with eng.guard():                       # request threads may share this module (app.py:50-61): one replay at a time
            feats = eng.backbone(patches, R)
            eng.head(feats, R, logits, labels)
        return logits


def build_network(CFG, modelName: Optional[str], weights: Optional[str]) -> Network_Wrapper:
    """Shared body of the two reference build_model()s (utils/utils.py:75-87, utils/trainClass.py:341-358):
    create backbone, wrap, strict load_state_dict(torch.load(path, map_location=CFG.device))."""
    name = modelName or CFG.modelName
    path = weights or CFG.pretrained
    net = Network_Wrapper(create_model(name, pretrained=False, num_classes=1000), CFG.num_classes)
    sd = torch.load(path, map_location=torch.device("cpu"), weights_only=True)
    net.load_state_dict(sd)
    return net
