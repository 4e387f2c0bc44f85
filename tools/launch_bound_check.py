#!/usr/bin/env python3
"""Dev tool (GPU box): is a training step bound by the host's launch rate?  Times the enqueue of N steps (no sync)
against the synchronised wall time, for the detector trainer and the ViT trainer."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
import yvhip
from yvhip.yolo_training import YoloTrainer, init_yolo_train_state

dev = "cuda:0"
scale, nc, S, B, G = "s", 80, 640, 16, 8
tr = YoloTrainer(init_yolo_train_state(scale, nc, seed=42), scale=scale, nc=nc, size=S, batch=B, device=dev)
g = torch.Generator().manual_seed(1)
images = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8).to(dev)
ctr = torch.rand(B, G, 2, generator=g) * S; wh = torch.rand(B, G, 2, generator=g) * 240 + 16
gtb = torch.cat([(ctr - wh / 2).clamp(0, S), (ctr + wh / 2).clamp(0, S)], -1).to(dev)
gtl = torch.randint(0, nc, (B, G), generator=g, dtype=torch.int32).to(dev)
gtn = torch.full((B,), G, dtype=torch.int32).to(dev)
for _ in range(3):
    tr.step(images, gtb, gtl, gtn)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    tr.step(images, gtb, gtl, gtn)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"yolo train: enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, wall {1e3 * (t2 - t0) / n:.2f} ms/step")
