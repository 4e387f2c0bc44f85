"""Isolated launches of the fused C2f kernel for tools/c2f_pmc.sh (C2F_C / C2F_N / C2F_H select the block)."""
import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "yolov8-vit_amd"))
import torch
import yvhip as yv
DEV = "cuda:0"
B, c, n, H = 32, int(os.environ.get("C2F_C", 32)), int(os.environ.get("C2F_N", 2)), int(os.environ.get("C2F_H", 80))
g = torch.Generator().manual_seed(0)
C1 = 2 * c
mk = lambda co, k: ((torch.randn(co, k, generator=g) * (2.0 / k) ** 0.5).to(torch.bfloat16).to(DEV), (torch.randn(co, generator=g) * 0.1).to(DEV))
w1, b1 = mk(C1, C1)
wm = [mk(c, 9 * c) for _ in range(2 * n)]
w2, b2 = mk(C1, (2 + n) * c)
x = torch.randn(B, H, H, C1, generator=g).to(torch.bfloat16).to(DEV)
out = torch.zeros(B, H, H, C1, dtype=torch.bfloat16, device=DEV)
for _ in range(6):
    yv.c2f_fused(x, c, n, w1, b1, [m[0] for m in wm], [m[1] for m in wm], w2, b2, out)
    torch.cuda.synchronize()
if os.environ.get("C2F_STAMPS"):
    dbg = torch.zeros(64 * 8, dtype=torch.int64, device=DEV)
    yv.lib.yv_c2f_debug(dbg.data_ptr())
    yv.c2f_fused(x, c, n, w1, b1, [m[0] for m in wm], [m[1] for m in wm], w2, b2, out)
    torch.cuda.synchronize()
    yv.lib.yv_c2f_debug(None)
    d = dbg.cpu().view(64, 8)
    dd = (d[:, 1:] - d[:, :-1]).double()
    print("stamp deltas (s_memtime ticks), mean over 64 workgroups:", [round(float(v)) for v in dd.mean(0)])
    print("first workgroup:", [int(v) for v in dd[0]])
