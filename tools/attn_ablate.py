import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "yolov8-vit_amd"))
import torch, ctypes
import yvhip
dev = "cuda:0"
R, N, H = int(os.environ.get("ATTN_R", 128)), 197, 12
g = torch.Generator().manual_seed(0)
qkv = torch.randn(R * N, 3 * H * 64, generator=g).to(torch.bfloat16).to(dev)
out = torch.zeros(R * N, H * 64, dtype=torch.bfloat16, device=dev)
big = torch.zeros(64 * 1024 * 1024, device=dev)
ref = None
for abl in (0, 5, 11, 12, 13, 0, 5):   # 11 / 12 / 13: the pipelined kernel without fetches / compute / output stores (timing only)   # 0 = shipped (round 3: pipelined items, LDS-DMA staging, transposing V reads), 5 = round-2 one-item kernel, 4 = round-1 form
    yvhip.lib.yv_attention_debug(abl)
    ts = []
    for rd in range(5):
        big.add_(1.0)                      # flush L2 / MALL a bit between rounds
        yvhip.attention(qkv, R, N, H, out); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            yvhip.attention(qkv, R, N, H, out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    o = out.float().clone()
    if ref is None: ref = o
    print(f"ablate={abl}: {sorted(ts)[2]:.1f} us  (R = {R}; max |diff| to the first variant {float((o - ref).abs().max()):.3g})")
yvhip.lib.yv_attention_debug(0)
