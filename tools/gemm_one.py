#!/usr/bin/env python3
"""Dev tool (GPU box, under rocprofv3 --pmc): one shape of the persistent GEMM, 10 launches (SHAPE=qkv|proj|fc1|fc2, GB_M rows)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
import yvhip
dev = "cuda:0"
M = int(os.environ.get("GB_M", 25216))
n, k, flags = {"qkv": (2304, 768, 0), "proj": (768, 768, yvhip.EPI_RES_F32), "fc1": (3072, 768, yvhip.EPI_GELU),
               "fc2": (768, 3072, yvhip.EPI_RES_F32)}[os.environ.get("SHAPE", "qkv")]
g = torch.Generator().manual_seed(0)
a = torch.randn(M, k, generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
bias = torch.randn(n, generator=g).to(dev)
out = torch.zeros(M, n, dtype=torch.float32 if flags & yvhip.EPI_RES_F32 else torch.bfloat16, device=dev)
yvhip.set_option("linear_variant", 9)
for _ in range(10):
    yvhip.linear(a, w, bias, out, flags=flags)
torch.cuda.synchronize()
