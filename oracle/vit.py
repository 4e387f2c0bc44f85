"""Oracle (TEST INFRASTRUCTURE): fp32 CPU forward of the classifier, rows
B3/B4 of SURVEY.md section 8(a).

Parity UNPINNED for B3: the arithmetic lives in `timm` (unpinned in
requirements.txt:5, call sites utils/utils.py:82, utils/trainClass.py:352),
absent from the reference tree and from this image.  The restatement follows
README.md:5-35 and timm's published VisionTransformer conventions (LN eps
1e-6, qkv bias, exact-erf GELU, scale d^-0.5, MLP ratio 4, cls token + learned
pos_embed, head on the cls token after the final norm).  B4 (Network_Wrapper,
utils/utils.py:59-72) is pinned by golden G6."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

VIT_CFGS = {
    # name: (patch, dim, depth, heads)
    "vit_base_patch16_224": (16, 768, 12, 12),
    "vit_base_patch8_224": (8, 768, 12, 12),
    "vit_large_patch16_224": (16, 1024, 24, 16),
    "vit_tiny_test": (16, 128, 2, 2),          # test-only miniature (d = 64)
    "vit_tiny8_test": (8, 128, 2, 2),          # test-only miniature with 785 tokens
}


def vit_cfg(name: str):
    base = name.split(".")[0]
    return VIT_CFGS[base]


def init_wrapper_state(name: str, num_classes: int = 5, seed: int = 42, img: int = 224,
                       std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Seeded random state_dict with the key layout of Network_Wrapper(timm
    ViT) (`model.*` + `fc.1.*`, `fc.3.*`; verified against the reference
    wrapper class by golden G6)."""
    P, D, L, H = vit_cfg(name)
    n = (img // P) ** 2 + 1
    g = torch.Generator().manual_seed(seed)

    def rn(*s, scale=std):
        return torch.randn(*s, generator=g) * scale

    sd = {
        "model.cls_token": rn(1, 1, D),
        "model.pos_embed": rn(1, n, D),
        "model.patch_embed.proj.weight": rn(D, 3, P, P),
        "model.patch_embed.proj.bias": rn(D),
    }
    for i in range(L):
        p = f"model.blocks.{i}."
        sd[p + "norm1.weight"] = 1 + rn(D)
        sd[p + "norm1.bias"] = rn(D)
        sd[p + "attn.qkv.weight"] = rn(3 * D, D, scale=0.04)
        sd[p + "attn.qkv.bias"] = rn(3 * D)
        sd[p + "attn.proj.weight"] = rn(D, D)
        sd[p + "attn.proj.bias"] = rn(D)
        sd[p + "norm2.weight"] = 1 + rn(D)
        sd[p + "norm2.bias"] = rn(D)
        sd[p + "mlp.fc1.weight"] = rn(4 * D, D)
        sd[p + "mlp.fc1.bias"] = rn(4 * D)
        sd[p + "mlp.fc2.weight"] = rn(D, 4 * D)
        sd[p + "mlp.fc2.bias"] = rn(D)
    sd["model.norm.weight"] = 1 + rn(D)
    sd["model.norm.bias"] = rn(D)
    sd["model.head.weight"] = rn(1000, D, scale=0.05)
    sd["model.head.bias"] = rn(1000, scale=0.5)
    sd["fc.1.weight"] = rn(128, 1000, scale=0.05)
    sd["fc.1.bias"] = rn(128, scale=0.1)
    sd["fc.3.weight"] = rn(num_classes, 128, scale=0.2)
    sd["fc.3.bias"] = rn(num_classes, scale=0.1)
    return sd


def backbone_param_count(sd: Dict[str, torch.Tensor]) -> int:
    return sum(v.numel() for k, v in sd.items() if k.startswith("model."))


def vit_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, name: str, return_tokens: bool = False):
    """x (R,3,H,W) f32 -> backbone logits (R,1000) f32.  README.md:9-35."""
    P, D, L, H = vit_cfg(name)
    d = D // H
    R = x.shape[0]
    t = F.conv2d(x, sd["model.patch_embed.proj.weight"], sd["model.patch_embed.proj.bias"], stride=P)
    t = t.flatten(2).transpose(1, 2)                                    # (R, n-1, D)
    t = torch.cat([sd["model.cls_token"].expand(R, -1, -1), t], dim=1) + sd["model.pos_embed"]
    N = t.shape[1]
    for i in range(L):
        p = f"model.blocks.{i}."
        h = F.layer_norm(t, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-6)
        qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
        qkv = qkv.reshape(R, N, 3, H, d).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = (q * (d ** -0.5)) @ k.transpose(-2, -1)
        att = att.softmax(dim=-1)
        o = (att @ v).transpose(1, 2).reshape(R, N, D)
        t = t + F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        h = F.layer_norm(t, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps=1e-6)
        h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        t = t + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    if return_tokens:
        return t
    c = F.layer_norm(t[:, 0], (D,), sd["model.norm.weight"], sd["model.norm.bias"], eps=1e-6)
    return F.linear(c, sd["model.head.weight"], sd["model.head.bias"])


def wrapper_head(sd: Dict[str, torch.Tensor], feats: torch.Tensor) -> torch.Tensor:
    """utils/utils.py:64-72: ReLU -> Linear(1000,128) -> ReLU -> Linear(128,nc)."""
    h = F.relu(feats)
    h = F.relu(F.linear(h, sd["fc.1.weight"], sd["fc.1.bias"]))
    return F.linear(h, sd["fc.3.weight"], sd["fc.3.bias"])


def wrapper_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, name: str) -> torch.Tensor:
    """Network_Wrapper.forward = fc(model(x)) (utils/utils.py:71-72)."""
    return wrapper_head(sd, vit_forward(sd, x, name))
