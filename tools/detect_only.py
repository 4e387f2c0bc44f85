#!/usr/bin/env python3
"""Dev tool (GPU box, under rocprofv3): the detect stage of the headline workload alone, 20 passes (kernel times without the
classifier's streams beside them)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
from yvhip import engines
from yvhip.pipeline import DetectClassifyPipeline
dev = "cuda:0"
name = "vit_base_patch16_224"
pipe = DetectClassifyPipeline(engines.YoloEngine(engines.init_yolo_state("n", 5, 42, 4.0), "n", 5, 640, dev),
                              [engines.VitEngine(engines.init_vit_wrapper_state(name, 5, 42), name, 5, device=dev)],
                              max_crops_per_image=4)
g = torch.Generator().manual_seed(1234)
images = torch.randint(0, 256, (32, 640, 640, 3), generator=g, dtype=torch.uint8).to(dev)
for _ in range(int(os.environ.get("PASSES", 20))):
    det = pipe.detect_stage(images)
torch.cuda.synchronize()
print("crops:", int(det["crop_total"]))
