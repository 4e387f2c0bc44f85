#!/usr/bin/env python3
"""Dev tool (GPU box): interleaved in-process A/B of whole-pipeline step time under different GEMM policies."""
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
runner = PipelinedRunner(pipe)
r_split = PipelinedRunner(pipe, split_classifier=True)
def with_opt(key, val, fn):
    def run(x):
        yvhip.set_option(key, val)
        return fn(x)
    return run


# E2E_OPTION="key:a,b" compares two values of a yv_set_option knob under the two-stream schedule
spec = os.environ.get("E2E_OPTION")
if spec:
    key, vals = spec.split(":")
    r_s = PipelinedRunner(pipe, split_classifier=True)          # the shipped schedule
    opts = [(f"{key}={v}", with_opt(key, int(v), r_s.submit)) for v in vals.split(",")]
elif os.environ.get("E2E_C2F"):
    r_s = PipelinedRunner(pipe, split_classifier=True)
    def with_c2f(flag):
        def run(x):
            pipe.yolo.fused_c2f = flag
            return r_s.submit(x)
        return run
    opts = [("C2f layer by layer", with_c2f(False)), ("C2f fused", with_c2f(True))]
elif os.environ.get("E2E_SPLIT"):
    opts = [("two streams", runner.submit)] + [(f"split x{k}", PipelinedRunner(pipe, split_classifier=int(k)).submit)
                                               for k in os.environ["E2E_SPLIT"].split(",")]
elif os.environ.get("E2E_GRAPH"):
    static = images.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        pipe(static)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gout = pipe(static)
    r_hi = PipelinedRunner(pipe, det_priority=-1)
    r_new = PipelinedRunner(pipe)
    opts = [("single stream eager", pipe), ("single stream hipGraph", lambda x: graph.replay()), ("two streams", runner.submit),
            ("two streams det-high-prio", r_hi.submit), ("two streams (new runner)", r_new.submit)]
elif os.environ.get("E2E_RUN_AHEAD"):
    opts = [(f"run_ahead={k}", PipelinedRunner(pipe, run_ahead=int(k)).submit) for k in os.environ["E2E_RUN_AHEAD"].split(",")]
else:
    opts = [("single stream", pipe), ("two streams", runner.submit), ("two streams + split ViT", r_split.submit)]
res = {k: [] for k, _ in opts}
for _ in range(3):
    pipe(images)
torch.cuda.synchronize()
for rd in range(6):
    for k, v in opts:
        v(images); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            v(images)
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 8 * 1e3)
for k, ts in res.items():
    ts = sorted(ts)
    print(f"{k:24s} median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f} ms  -> {32/ts[len(ts)//2]*1e3:.0f} img/s")
