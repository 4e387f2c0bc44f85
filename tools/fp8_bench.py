#!/usr/bin/env python3
"""Dev tool (GPU box): MXFP8 block-scaled GEMM vs the bf16 LDS-DMA GEMM on the classifier shapes (interleaved)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
import yvhip
dev = "cuda:0"
which = os.environ.get("FB_MODEL", "L")
M = int(os.environ.get("FB_M", 50432 if which == "L" else 25216))
D = 1024 if which == "L" else 768
shapes = [("qkv", M, 3 * D, D, 0), ("proj", M, D, D, yvhip.EPI_RES_F32), ("fc1", M, 4 * D, D, yvhip.EPI_GELU), ("fc2", M, D, 4 * D, yvhip.EPI_RES_F32)]
g = torch.Generator().manual_seed(0)
for name, m, n, k, flags in shapes:
    a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    pad = int(os.environ.get("FB_PAD", "0"))                  # row-stride padding in bytes (channel-camping experiment)
    aq, asc = yvhip.quant_mxfp8(a)
    if pad:                                                    # strided copies, passed to the C ABI directly
        a_p = torch.zeros(m, k + pad // 2, dtype=a.dtype, device=dev); a_p[:, :k] = a
        aq_p = torch.zeros(m, k + pad, dtype=torch.uint8, device=dev); aq_p[:, :k] = aq
    wq, wsc = yvhip.quant_mxfp8(w)
    out = torch.zeros(m, n, dtype=torch.float32 if flags & yvhip.EPI_RES_F32 else torch.bfloat16, device=dev)
    res = {"bf16": [], "mxfp8": [], "quant_a": []}
    for rd in range(5):
        for key in res:
            P = yvhip._p
            if pad:
                fb = lambda: yvhip.check(yvhip.lib.yv_linear(P(a_p), a_p.stride(0), P(w), P(bias), m, n, k, P(out), out.stride(0), None, 0,
                                                              flags | yvhip.EPI_BIAS, None, 1, yvhip._st()), "yv_linear")
                fm = lambda: yvhip.check(yvhip.lib.yv_linear_mxfp8(P(aq_p), aq_p.stride(0), P(asc), asc.shape[1], P(wq), P(wsc), wsc.shape[1],
                                                                    P(bias), m, n, k, P(out), out.stride(0), flags | yvhip.EPI_BIAS, None, 1,
                                                                    yvhip._st()), "yv_linear_mxfp8")
            else:
                fb = lambda: yvhip.linear(a, w, bias, out, flags=flags)
                fm = lambda: yvhip.linear_mxfp8(aq, asc, wq, wsc, bias, out, flags=flags)
            fn = {"bf16": fb,
                  "mxfp8": fm,
                  "quant_a": lambda: yvhip.quant_mxfp8(a, aq, asc)}[key]
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            res[key].append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * m * n * k
    line = f"{name:5s} M={m} N={n} K={k}:"
    for key, ts in res.items():
        t = sorted(ts)[len(ts) // 2]
        line += f"  {key} {t*1e3:7.1f} us" + (f" ({fl/t/1e9:6.0f} TF)" if key != "quant_a" else "")
    print(line, flush=True)
