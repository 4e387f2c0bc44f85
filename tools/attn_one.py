#!/usr/bin/env python3
"""Dev tool (GPU box, under rocprofv3 --pmc): the 197-token attention kernel on 128 crops x 12 heads, 10 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
import yvhip
dev = "cuda:0"
R, N, H = int(os.environ.get("CROPS", 128)), 197, 12
g = torch.Generator().manual_seed(0)
qkv = torch.randn(R * N, 3 * H * 64, generator=g).to(torch.bfloat16).to(dev)
out = torch.zeros(R * N, H * 64, dtype=torch.bfloat16, device=dev)
for _ in range(10):
    yvhip.attention(qkv, R, N, H, out)
torch.cuda.synchronize()
