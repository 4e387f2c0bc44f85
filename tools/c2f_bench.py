"""Backbone C2f blocks of YOLOv8n at batch 32: one fused launch (yv_c2f_fused) against the layer-by-layer path (yv_conv2d x 4 / 6)."""
import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "yolov8-vit_amd"))
import torch
import yvhip as yv
DEV = "cuda:0"
B = int(os.environ.get("C2F_B", 32))
g = torch.Generator().manual_seed(0)
flush = torch.zeros(64 * 1024 * 1024, device=DEV)


def timed(fn, reps=10, rounds=5):
    ts = []
    for _ in range(rounds):
        flush.add_(1.0)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[rounds // 2]


for c, n, H in ((16, 1, 160), (32, 2, 80)):
    C1 = 2 * c
    mk = lambda co, k: ((torch.randn(co, k, generator=g) * (2.0 / k) ** 0.5).to(torch.bfloat16).to(DEV), (torch.randn(co, generator=g) * 0.1).to(DEV))
    w1, b1 = mk(C1, C1)
    wm = [mk(c, 9 * c) for _ in range(2 * n)]
    w2, b2 = mk(C1, (2 + n) * c)
    x = torch.randn(B, H, H, C1, generator=g).to(torch.bfloat16).to(DEV)
    y = torch.zeros(B, H, H, (2 + n) * c, dtype=torch.bfloat16, device=DEV)
    t = torch.zeros(B, H, H, c, dtype=torch.bfloat16, device=DEV)
    ref = torch.zeros(B, H, H, C1, dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(B, H, H, C1, dtype=torch.bfloat16, device=DEV)

    def layers():
        yv.conv2d(yv.view(x, 0, C1), None, B, H, H, 1, 1, w1, b1, y, 0, yv.EPI_SILU)
        for j in range(n):
            src = (1 + j) * c
            yv.conv2d(yv.view(y, src, c), None, B, H, H, 3, 1, wm[2 * j][0], wm[2 * j][1], t, 0, yv.EPI_SILU)
            yv.conv2d(yv.view(t, 0, c), None, B, H, H, 3, 1, wm[2 * j + 1][0], wm[2 * j + 1][1], y, src + c,
                      yv.EPI_SILU | yv.EPI_RES_BF16, res=y, res_c_off=src)
        yv.conv2d(yv.view(y, 0, (2 + n) * c), None, B, H, H, 1, 1, w2, b2, ref, 0, yv.EPI_SILU)

    fused = lambda: yv.c2f_fused(x, c, n, w1, b1, [m[0] for m in wm], [m[1] for m in wm], w2, b2, out)
    tl, tf = timed(layers), timed(fused)
    mb = 2 * B * H * H * C1 * 2 / 1e6
    print(f"C2f c={c} n={n} {B} x {H} x {H} x {C1}: layer by layer {tl:7.1f} us   fused {tf:7.1f} us   ({mb:.0f} MB in + out = "
          f"{mb / tf:.2f} TB/s; equal: {bool(torch.equal(out, ref))})")
