#!/usr/bin/env python3
"""Dev tool (GPU box): device letterbox (bilinear, yv_letterbox) of a batch of camera-sized frames into 640 x 640."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
import yvhip
dev = "cuda:0"
for (B, H, W) in ((32, 1080, 1920), (32, 720, 1280), (32, 480, 640)):
    g = torch.Generator().manual_seed(1)
    src = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8).to(dev)
    r = min(640 / H, 640 / W)
    nw, nh = int(round(W * r)), int(round(H * r))
    left, top = (640 - nw) // 2, (640 - nh) // 2
    geom = torch.tensor([[W, H, nw, nh, left, top]] * B, dtype=torch.int32, device=dev)
    for _ in range(3):
        out = yvhip.letterbox(src, geom, 640)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = yvhip.letterbox(src, geom, 640)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{B} x {H}x{W} -> 640: {us:.0f} us  ({B * 640 * 640 * 3 / us / 1e3:.2f} GB/s of output)")
