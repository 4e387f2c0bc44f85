"""Classify stage alone (128 crops, one stream) on CU-masked streams: mask bit i = XCC i % 8, CU i / 8 of that XCC
(tools/cu_mask_probe.hip); an XCC without any bit set is unrestricted."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
import yvhip
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
det = pipe.detect_stage(images)
torch.cuda.synchronize()
for _ in range(2):
    pipe.classify_stage(images, det)
torch.cuda.synchronize()


def run(stream, cus, n=6):
    yvhip.set_option("linear_p8_cus", cus)
    with torch.cuda.stream(stream):
        pipe.classify_stage(images, det)
    stream.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(n):
            pipe.classify_stage(images, det)
    stream.synchronize()
    yvhip.set_option("linear_p8_cus", 0)
    return (time.perf_counter() - t0) / n * 1e3


print(f"plain stream, GEMM grid 256:                     {run(torch.cuda.Stream(), 0):.3f} ms")
print(f"plain stream, GEMM grid 208:                     {run(torch.cuda.Stream(), 208):.3f} ms")
print(f"masked stream, all 256 bits, GEMM grid 256:      {run(masked_stream(range(256)), 0):.3f} ms")
print(f"masked stream, bits 48..255 (26 per XCC), 208:   {run(masked_stream(range(48, 256)), 208):.3f} ms")
print(f"masked stream, bits 64..255 (24 per XCC), 192:   {run(masked_stream(range(64, 256)), 192):.3f} ms")
print(f"masked stream, bits 48..255, GEMM grid 256:      {run(masked_stream(range(48, 256)), 0):.3f} ms")
