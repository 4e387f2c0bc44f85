#!/usr/bin/env python3
"""Dev tool (GPU box): time yv_linear variants on the ViT-B/16 shapes of the bench, interleaved in one
process, and check them against torch.matmul (hipBLASLt: comparison baseline only, never the product)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
import yvhip

dev = "cuda:0"
M = int(os.environ.get("GB_M", 25216))
shapes = [("qkv", M, 2304, 768), ("proj", M, 768, 768), ("fc1", M, 3072, 768), ("fc2", M, 768, 3072)]
variants = [int(v) for v in os.environ.get("GB_VARIANTS", "1,3,10").split(",")]
groups = [int(v) for v in os.environ.get("GB_GROUPS", "8").split(",")]
rounds = 5
g = torch.Generator().manual_seed(0)
for name, m, n, k in shapes:
    a = (torch.randn(m, k, generator=g)).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    epi = os.environ.get("GB_EPI", "plain")
    flags, odt = 0, torch.bfloat16
    if epi == "pipe":
        flags = {"qkv": 0, "proj": yvhip.EPI_RES_F32, "fc1": yvhip.EPI_GELU, "fc2": yvhip.EPI_RES_F32}[name]
        odt = torch.float32 if flags & yvhip.EPI_RES_F32 else torch.bfloat16
    out = torch.zeros(m, n, dtype=odt, device=dev)
    ref = (a @ w.t()).float() + bias
    if flags & yvhip.EPI_GELU:
        ref = torch.nn.functional.gelu(ref)
    res = {}
    for rd in range(rounds):
        for v in variants:
            for gm in groups:
                yvhip.set_option("linear_variant", 9 if v >= 900 else v); yvhip.set_option("linear_group_m", gm)
                yvhip.set_option("linear_p8_rows", v - 900 if v >= 900 else 0)      # 9xx: persistent kernel with xx0.. rows forced
                yvhip.set_option("linear_p8_sched", 1 if v == 91 else 0)
                if v == 91:
                    yvhip.set_option("linear_variant", 9)                            # 91: persistent kernel, XCD-contiguous schedule
                out.zero_()
                yvhip.linear(a, w, bias, out, flags=flags)
                torch.cuda.synchronize()
                first = out.float().clone() if rd == 0 else None
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    yvhip.linear(a, w, bias, out, flags=flags)
                e1.record(); torch.cuda.synchronize()
                res.setdefault((v, gm), []).append(e0.elapsed_time(e1) / 10)
                if rd == 0:
                    err = float((first - ref).norm() / ref.norm())
                    assert err < 5e-3 or v > 100, (name, v, err)
                    if os.environ.get("GB_PRINT_ERR"):
                        print(f"   {name} variant {v}: rel-L2 vs torch {err:.2e}")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.matmul(a, w.t())
        e1.record(); torch.cuda.synchronize()
        res.setdefault(("hipblaslt", 0), []).append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * m * n * k
    for key, ts in res.items():
        ts = sorted(ts)
        print(f"{name:5s} M={m} N={n} K={k} variant={key[0]} gm={key[1]}: median {ts[len(ts)//2]*1e3:8.1f} us  "
              f"{fl/ts[len(ts)//2]/1e9:7.1f} TF/s (min {fl/ts[0]/1e9:7.1f})", flush=True)
