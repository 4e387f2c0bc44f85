"""Detect stage alone and the whole pipelined step under the convolution schedules of yv_set_option("conv_dma", v), interleaved."""
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
vals = [int(v) for v in os.environ.get("CONV_DMA", "8,2,1,7").split(",")]
res = {(v, k): [] for v in vals for k in ("detect", "step")}
for _ in range(3):
    pipe(images)
torch.cuda.synchronize()
for rd in range(6):
    for v in vals:
        yvhip.set_option("conv_dma", v)
        pipe.detect_stage(images); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            pipe.detect_stage(images)
        torch.cuda.synchronize()
        res[(v, "detect")].append((time.perf_counter() - t0) / 8 * 1e3)
        runner.submit(images); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            runner.submit(images)
        torch.cuda.synchronize()
        res[(v, "step")].append((time.perf_counter() - t0) / 8 * 1e3)
yvhip.set_option("conv_dma", 8)
for v in vals:
    d, s = sorted(res[(v, "detect")]), sorted(res[(v, "step")])
    print(f"conv_dma={v}: detect stage alone median {d[3]:.3f} ms (min {d[0]:.3f})   pipelined step median {s[3]:.3f} ms (min {s[0]:.3f})")
