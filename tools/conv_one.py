#!/usr/bin/env python3
"""Dev tool (GPU box, under rocprofv3 --pmc): one convolution launch of the YOLOv8n forward (LAYER = index in
tools/conv_layers.py's table) repeated 10 times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
from yvhip import engines
dev = "cuda:0"
eng = engines.YoloEngine(engines.init_yolo_state("n", 5, 42, 4.0), "n", 5, 640, dev)
g = torch.Generator().manual_seed(1234)
images = torch.randint(0, 256, (32, 640, 640, 3), generator=g, dtype=torch.uint8).to(dev)
calls = []
orig = engines.conv2d
def rec(*a, **k):
    calls.append((a, k)); return orig(*a, **k)
engines.conv2d = rec
eng(images); torch.cuda.synchronize()
engines.conv2d = orig
a, k = calls[int(os.environ.get("LAYER", 44))]
torch.cuda.synchronize()
for _ in range(10):
    orig(*a, **k)
torch.cuda.synchronize()
