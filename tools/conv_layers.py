#!/usr/bin/env python3
"""Dev tool (GPU box): every convolution launch of one YOLOv8 forward, replayed alone (20 back-to-back launches between HIP
events): implicit-GEMM shape, microseconds, TFLOP/s and GB/s against its algorithmic bytes (input + weights + output once)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
import yvhip
from yvhip import engines
dev = "cuda:0"
scale, nc, S, B = os.environ.get("SCALE", "n"), 5, 640, int(os.environ.get("BATCH", 32))
eng = engines.YoloEngine(engines.init_yolo_state(scale, nc, 42, 4.0), scale, nc, S, dev)
g = torch.Generator().manual_seed(1234)
images = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8).to(dev)
calls = []
orig = engines.conv2d
def rec(*a, **k):
    calls.append((a, k))
    return orig(*a, **k)
engines.conv2d = rec
eng(images); torch.cuda.synchronize()
engines.conv2d = orig
tot = 0.0
print(f"{'#':>3} {'M':>8} {'Cout':>5} {'K':>5} {'k':>2} {'s':>2} {'in1':>4} {'us':>8} {'TF/s':>7} {'GB/s':>7}  floor_us(8TB/s)")
for n, (a, k) in enumerate(calls):
    in0, in1, Bb, Ho, Wo, ks, st, w, bias, out, off, flags = a[:12]
    cin = in0.c + (in1.c if in1 is not None else 0)
    M, N, K = Bb * Ho * Wo, w.shape[0], cin * ks * ks
    for _ in range(3):
        orig(*a, **k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        orig(*a, **k)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    esz = 4 if flags & yvhip.EPI_OUT_F32 else 2
    up = 2 if (in0.up or (in1 is not None and in1.up)) else 1
    byt = Bb * (Ho * st) * (Wo * st) * cin * 2 / (up * up if in0.up else 1) + N * K * 2 + M * N * esz
    tot += us
    print(f"{n:3d} {M:8d} {N:5d} {K:5d} {ks:2d} {st:2d} {('y' if in1 is not None else '-'):>4} {us:8.1f} {2.0*M*N*K/us/1e6:7.1f} {byt/us/1e3:7.0f}  {byt/8e6:6.1f}")
print("sum of isolated conv times: %.1f us" % tot)
