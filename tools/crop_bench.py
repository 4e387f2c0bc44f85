"""crop_kernel<2> at B x 4 crops (SURVEY 8(d) synthetic boxes) by row groups per block (yv_crop_debug; 0 = chosen by the launcher)."""
import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "yolov8-vit_amd"))
import torch
import yvhip
dev = "cuda:0"
for B, wide in ((32, False), (256, False), (256, True)):
    A, S, R = 8400, 640, 4
    g = torch.Generator().manual_seed(4321)
    ctr = torch.rand(B, A, 2, generator=g) * S
    wh = torch.rand(B, A, 2, generator=g) * 240 + 16
    if wide:
        wh = torch.rand(B, A, 2, generator=g) * 300 + 340       # 340 .. 640 pixel boxes: two or four staging passes per row group
    boxes = torch.cat([(ctr - wh / 2).clamp(0, S), (ctr + wh / 2).clamp(0, S)], -1)
    images = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8).to(dev)
    cl = torch.zeros(B * R, 6, dtype=torch.int32)
    for b in range(B):
        for k in range(R):
            x0, y0, x1, y1 = [int(v) for v in boxes[b, k].tolist()]
            cl[b * R + k] = torch.tensor([b, x0, y0, max(x1, x0 + 8), max(y1, y0 + 8), k])
    cl = cl.to(dev)
    total = torch.tensor([B * R], dtype=torch.int32, device=dev)
    out = torch.zeros((B * R * 196, 768), dtype=torch.bfloat16, device=dev)
    ref = None
    for gpb in ((0, 1, 2) if wide else (1, 0, 2, 4, 14)):
        yvhip.lib.yv_crop_debug(gpb)
        out.zero_()
        for _ in range(3):
            yvhip.crop_resize_norm(images, cl, total, B * R, 224, 16, layout=2, out=out)
        torch.cuda.synchronize()
        ts = []
        for rd in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                yvhip.crop_resize_norm(images, cl, total, B * R, 224, 16, layout=2, out=out)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        t = sorted(ts)[2]
        if ref is None: ref = out.clone()
        print(f"{B * R:5d} {'wide ' if wide else ''}crops, groups per block {gpb:2d}: {t:7.1f} us  {B * R * 451584 / t * 1e-6:6.2f} TB/s ({B * R * 451584 / t * 1e-6 / 8 * 100:.0f} % of 8 TB/s)  identical: {bool(torch.equal(out, ref))}")
yvhip.lib.yv_crop_debug(0)
