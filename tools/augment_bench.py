"""Times the two augmentation gather kernels against their HBM roofline (algorithmic bytes / time).
    python tools/augment_bench.py            (on the GPU box)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import yvhip                                                   # noqa: E402
from yvhip.augment import TrainAugment                         # noqa: E402
from yvhip.yolo_augment import DetAugment, build_record, tile_geometry   # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3                         # us


def main():
    yvhip.require_gpu()
    S, P, B = 224, 16, 256                                     # classifier crops: (B,3,S,S) f32 in, bf16 patch rows out
    x = torch.randn(B, 3, S, S, device=DEV)
    geo, idx = TrainAugment(S, seed=1).sample(B)
    geo, idx = torch.from_numpy(geo).to(DEV), torch.from_numpy(idx).to(DEV)
    out = torch.empty((B * (S // P) ** 2, 3 * P * P), dtype=torch.bfloat16, device=DEV)
    us = timeit(lambda: yvhip.augment_patchify(x, geo, idx, P, out))
    alg = B * 3 * S * S * (4 + 2)
    print(f"augment_patchify  B={B} S={S}: {us:8.1f} us  {alg / us / 1e3:7.1f} GB/s algorithmic ({alg / 1e6:.1f} MB)")
    from yvhip.modules import patchify_bf16
    us0 = timeit(lambda: patchify_bf16(x, P))
    print(f"  torch patchify_bf16 (no augmentation) for scale: {us0:8.1f} us")

    S, B, n_tiles = 640, 16, 40                                # detector inputs: 4 tiles per output image
    rng = np.random.default_rng(0)
    sizes = [tile_geometry(int(rng.integers(300, 1400)), int(rng.integers(300, 1400)), S) for _ in range(n_tiles)]
    tiles = torch.randint(0, 256, (n_tiles, S, S, 3), dtype=torch.uint8, device=DEV)
    aug = DetAugment(S, seed=2)
    rf, ri, lut = [], [], []
    for b in range(B):
        p = aug.plan(b, n_tiles)
        f, i, l, _, _, _ = build_record(p, [sizes[s] for s in p["sources"]], p["sources"], S)
        rf.append(f); ri.append(i); lut.append(l)
    rf, ri, lut = (torch.from_numpy(np.stack(a)).to(DEV) for a in (rf, ri, lut))
    us = timeit(lambda: yvhip.mosaic_augment(tiles, rf, ri, lut))
    alg = B * S * S * 3 * 2                                    # one source byte (at scale 1) + one output byte per value
    print(f"mosaic_augment    B={B} S={S}: {us:8.1f} us  {alg / us / 1e3:7.1f} GB/s algorithmic ({alg / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
