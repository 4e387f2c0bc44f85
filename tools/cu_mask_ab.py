"""Experiment: give the detector stream its own CUs (hipExtStreamCreateWithCUMask) instead of letting its kernels squeeze in
between the persistent GEMM workgroups.  DET_CUS = CUs for the detector; LAYOUT = 'rr' (mask bits i with i % stride == 0 ...) or
'block' (the first DET_CUS bits)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
import yvhip
from yvhip import engines
from yvhip.pipeline import DetectClassifyPipeline, PipelinedRunner
dev = "cuda:0"
torch.cuda.init()
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = [0] * 8
    for b in bits:
        words[b >> 5] |= 1 << (b & 31)
    arr = (ctypes.c_uint32 * 8)(*words)
    s = ctypes.c_void_p()
    err = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, arr)
    if err != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {err}")
    return torch.cuda.ExternalStream(s.value)


name = "vit_base_patch16_224"
pipe = DetectClassifyPipeline(engines.YoloEngine(engines.init_yolo_state("n", 5, 42, 4.0), "n", 5, 640, dev),
                              [engines.VitEngine(engines.init_vit_wrapper_state(name, 5, 42), name, 5, device=dev)],
                              max_crops_per_image=4)
g = torch.Generator().manual_seed(1234)
images = torch.randint(0, 256, (32, 640, 640, 3), generator=g, dtype=torch.uint8).to(dev)
base = PipelinedRunner(pipe, split_classifier=True)
variants = [("shipped (no masks)", base)]
for spec in os.environ.get("SPECS", "48:block,48:rr,64:block,64:rr,32:rr").split(","):
    k, layout = spec.split(":")
    k = int(k)
    if layout == "block":
        det_bits = list(range(k))
    else:                                  # every (256 / k)-th CU
        det_bits = sorted({int(i * 256 / k) for i in range(k)})
    cls_bits = [b for b in range(256) if b not in set(det_bits)]
    if layout == "detonly":                # bit i = XCC i % 8, CU i / 8 (tools/cu_mask_probe.hip): the first k / 8 CUs of every XCC
        det_bits = list(range(k))
        r = PipelinedRunner(pipe, split_classifier=True)
        r.s_det = masked_stream(det_bits)
        variants.append((f"detector masked to {k} CUs, classifier unmasked", r))
        continue
    r = PipelinedRunner(pipe, split_classifier=True, gemm_cus=len(cls_bits))
    r.s_det = masked_stream(det_bits)
    r.s_cls = masked_stream(cls_bits)
    r.s_sub = [masked_stream(cls_bits) for _ in r.s_sub]
    variants.append((f"detector on {k} CUs ({layout}), classifier on {len(cls_bits)}", r))
res = {k: [] for k, _ in variants}
for _ in range(3):
    pipe(images)
torch.cuda.synchronize()
for rd in range(5):
    for k, r in variants:
        r.submit(images); r.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            r.submit(images)
        r.sync(); torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 8 * 1e3)
for k, ts in res.items():
    ts = sorted(ts)
    print(f"{k:50s} median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f} ms  -> {32 / ts[len(ts) // 2] * 1e3:.0f} img/s")
