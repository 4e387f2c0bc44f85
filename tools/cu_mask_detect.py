"""How long the detect stage takes when its stream is confined to k CUs (hipExtStreamCreateWithCUMask, every (256/k)-th mask bit),
alone on the chip - is "detector on the CUs the persistent GEMMs leave free" viable at all?"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
from yvhip import engines
from yvhip.pipeline import DetectClassifyPipeline
dev = "cuda:0"
torch.cuda.init()
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = [0] * 8
    for b in bits:
        words[b >> 5] |= 1 << (b & 31)
    s = ctypes.c_void_p()
    err = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, (ctypes.c_uint32 * 8)(*words))
    if err != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {err}")
    return torch.cuda.ExternalStream(s.value)


name = "vit_base_patch16_224"
pipe = DetectClassifyPipeline(engines.YoloEngine(engines.init_yolo_state("n", 5, 42, 4.0), "n", 5, 640, dev),
                              [engines.VitEngine(engines.init_vit_wrapper_state(name, 5, 42), name, 5, device=dev)],
                              max_crops_per_image=4)
g = torch.Generator().manual_seed(1234)
images = torch.randint(0, 256, (32, 640, 640, 3), generator=g, dtype=torch.uint8).to(dev)
for _ in range(3):
    pipe.detect_stage(images)
torch.cuda.synchronize()


def run(stream, n=8):
    with torch.cuda.stream(stream):
        pipe.detect_stage(images)
    stream.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(n):
            pipe.detect_stage(images)
    stream.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"plain stream, all CUs:        {run(torch.cuda.Stream()):.3f} ms")
for k in (256, 128, 64, 48, 32):
    bits = sorted({int(i * 256 / k) for i in range(k)})
    print(f"masked stream, {k:3d} CUs (every {256 // k}th bit): {run(masked_stream(bits)):.3f} ms")
bits = list(range(64))
print(f"masked stream,  64 CUs (bits 0..63):        {run(masked_stream(bits)):.3f} ms")
