import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "yolov8-vit_amd"))
import torch, ctypes
import yvhip
dev = "cuda:0"
R, N, H = 128, 197, 12
g = torch.Generator().manual_seed(0)
qkv = torch.randn(R * N, 3 * H * 64, generator=g).to(torch.bfloat16).to(dev)
out = torch.zeros(R * N, H * 64, dtype=torch.bfloat16, device=dev)
big = torch.zeros(64 * 1024 * 1024, device=dev)
for abl in (0, 4, 1, 2, 3, 0, 4):   # 0 = shipped (online softmax per 32-key group), 4 = round-1 form, 1-3 = ablations of the round-1 form
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
    print(f"ablate={abl}: {sorted(ts)[2]:.1f} us")
yvhip.lib.yv_attention_debug(0)
