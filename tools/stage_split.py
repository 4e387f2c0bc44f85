#!/usr/bin/env python3
"""Dev tool (GPU box): where does a headline step go?  Times, in one process and interleaved: the detect stage alone, the
classify stage alone (one stream, and the two-half split on two streams) on a fixed crop list, and the pipelined whole."""
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
runner = PipelinedRunner(pipe, split_classifier=True)      # gemm_cus = 208 around its classifier submissions
det = pipe.detect_stage(images)
torch.cuda.synchronize()
print("crops:", int(det["crop_total"]))
subs = [torch.cuda.Stream(), torch.cuda.Stream()]

def full(cus):
    def run():
        runner.gemm_cus = cus
        runner.submit(images)
    return run

def cls(streams, cus):
    def run():
        yvhip.set_option("linear_p8_cus", cus)
        pipe.classify_stage(images, det, streams)
    return run

opts = [("detect stage alone", lambda: pipe.detect_stage(images)),
        ("classify alone, 1 stream, 256 CUs", cls(None, 0)),
        ("classify alone, 1 stream, 208 CUs", cls(None, 208)),
        ("classify alone, split, 256 CUs", cls(subs, 0)),
        ("classify alone, split, 208 CUs", cls(subs, 208)),
        ("pipelined whole, 208 CUs", full(208))]
res = {k: [] for k, _ in opts}
for rd in range(5):
    for k, fn in opts:
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 8 * 1e3)
for k, ts in res.items():
    ts = sorted(ts)
    print(f"{k:36s} median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f} ms")
