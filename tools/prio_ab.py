"""Experiment: stream priorities of the two half-batch classifier streams (the detector stream is high priority in the shipped
schedule, both classifier streams normal)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
from yvhip import engines
from yvhip.pipeline import DetectClassifyPipeline, PipelinedRunner
dev = "cuda:0"
name = "vit_base_patch16_224"
pipe = DetectClassifyPipeline(engines.YoloEngine(engines.init_yolo_state("n", 5, 42, 4.0), "n", 5, 640, dev),
                              [engines.VitEngine(engines.init_vit_wrapper_state(name, 5, 42), name, 5, device=dev)],
                              max_crops_per_image=4)
g = torch.Generator().manual_seed(1234)
images = torch.randint(0, 256, (32, 640, 640, 3), generator=g, dtype=torch.uint8).to(dev)


def runner(det_prio, sub_prios):
    r = PipelinedRunner(pipe, split_classifier=True, det_priority=det_prio)
    r.s_sub = [torch.cuda.Stream(priority=p) for p in sub_prios]
    return r


variants = [("shipped: det high, halves normal / normal", runner(-1, (0, 0))),
            ("det high, halves high / normal", runner(-1, (-1, 0))),
            ("det normal, halves high / normal", runner(0, (-1, 0))),
            ("det normal, halves normal / normal", runner(0, (0, 0))),
            ("det high, halves high / high", runner(-1, (-1, -1)))]
res = {k: [] for k, _ in variants}
for _ in range(3):
    pipe(images)
torch.cuda.synchronize()
for rd in range(6):
    for k, r in variants:
        r.submit(images); r.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            r.submit(images)
        r.sync(); torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 8 * 1e3)
for k, ts in res.items():
    ts = sorted(ts)
    print(f"{k:44s} median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f} ms  -> {32 / ts[len(ts) // 2] * 1e3:.0f} img/s")
