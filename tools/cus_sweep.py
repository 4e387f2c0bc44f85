#!/usr/bin/env python3
"""Dev tool (GPU box): pipelined whole step against the number of CUs the persistent classifier GEMMs take, interleaved in one
process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
import yvhip
from yvhip import engines
from yvhip.pipeline import DetectClassifyPipeline, PipelinedRunner
dev = "cuda:0"
name = "vit_base_patch16_224"
pipe = DetectClassifyPipeline(engines.YoloEngine(engines.init_yolo_state("n", 5, 42, 4.0), "n", 5, 640, dev),
                              [engines.VitEngine(engines.init_vit_wrapper_state(name, 5, 42), name, 5, device=dev)],
                              max_crops_per_image=4)
g = torch.Generator().manual_seed(1234)
images = torch.randint(0, 256, (32, 640, 640, 3), generator=g, dtype=torch.uint8).to(dev)
runner = PipelinedRunner(pipe, split_classifier=True)
cus = [int(c) for c in os.environ.get("CUS", "176,192,200,208,216,224,240,256").split(",")]
if os.environ.get("FULL_FROM"):        # sweep the block from which the GEMMs take every CU again (gemm_cus fixed at its default)
    cus = [int(c) for c in os.environ["FULL_FROM"].split(",")]
res = {c: [] for c in cus}
for _ in range(3):
    runner.submit(images)
torch.cuda.synchronize()
for rd in range(5):
    for c in cus:
        if os.environ.get("FULL_FROM"):
            runner.full_cus_from = None if c < 0 else c
        else:
            runner.gemm_cus = c
        runner.submit(images); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            runner.submit(images)
        torch.cuda.synchronize()
        res[c].append((time.perf_counter() - t0) / 10 * 1e3)
for c, ts in res.items():
    ts = sorted(ts)
    print(f"{'full CUs from block' if os.environ.get('FULL_FROM') else 'gemm CUs'} {c:3d}: median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f} ms")
