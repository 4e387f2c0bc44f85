#!/usr/bin/env python3
"""Race screen for the 8-phase GEMM (GPU box): its result must be BIT-IDENTICAL to the simple 2-stage 256x256
kernel (same per-element accumulation order), on every repetition, for full, ragged and tiny shapes, with the
consumer L1/L2 warm and under back-to-back launches.  Any LDS read-before-landed / restage-before-read race
shows up as a mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
import yvhip

dev = "cuda:0"
shapes = [(25216, 2304, 768), (25216, 3072, 768), (25216, 768, 3072), (6304, 2304, 768), (6304, 768, 768),
          (1000, 512, 128), (256, 256, 64), (300, 256, 192), (257, 264, 64), (5000, 1024, 1024), (197, 2304, 768)]
reps = int(os.environ.get("REPS", 25))
g = torch.Generator().manual_seed(0)
bad = 0
for (m, n, k) in shapes:
    a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    for flags, dt in ((0, torch.bfloat16), (yvhip.EPI_GELU, torch.bfloat16), (yvhip.EPI_OUT_F32, torch.float32)):
        ref = torch.zeros(m, n, dtype=dt, device=dev)
        yvhip.set_option("linear_variant", 3)
        yvhip.linear(a, w, bias, ref, flags=flags)
        chk = (a.float() @ w.float().t() + bias)
        if flags & yvhip.EPI_GELU:
            chk = torch.nn.functional.gelu(chk)
        err = float((ref.float() - chk).norm() / chk.norm())
        assert err < 5e-3, (m, n, k, err)
        yvhip.set_option("linear_variant", 8)
        mism = 0
        for r in range(reps):
            out = torch.full((m, n), 3.0, dtype=dt, device=dev)
            yvhip.linear(a, w, bias, out, flags=flags)
            if not torch.equal(out, ref):
                mism += 1
        print(f"M={m} N={n} K={k} flags={flags}: {reps - mism}/{reps} identical", flush=True)
        bad += mism
yvhip.set_option("linear_variant", 1)
print("RACE SCREEN", "FAILED" if bad else "PASSED", bad)
sys.exit(1 if bad else 0)
