"""Isolated launches of the attention backward kernels (ViT-B/16 fine-tune shapes: 32 crops x 12 heads x 197 tokens) for PMC runs."""
import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "yolov8-vit_amd"))
import torch
import yvhip as yv
dev = "cuda:0"
R, N, H = int(os.environ.get("ATTN_R", 32)), 197, 12
D = H * 64
g = torch.Generator().manual_seed(0)
qkv = torch.randn(R * N, 3 * D, generator=g).to(torch.bfloat16).to(dev)
out = torch.zeros(R * N, D, dtype=torch.bfloat16, device=dev)
lse = torch.zeros(R * H * N, dtype=torch.float32, device=dev)
dout = torch.randn(R * N, D, generator=g).to(torch.bfloat16).to(dev)
dqkv = torch.zeros(R * N, 3 * D, dtype=torch.bfloat16, device=dev)
delta = torch.zeros(R * H * N + 64, dtype=torch.float32, device=dev)
yv.attention_train(qkv, R, N, H, out, lse)
import time
for _ in range(3):
    yv.attention_bwd(qkv, out, dout, lse, R, N, H, dqkv, delta)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    yv.attention_bwd(qkv, out, dout, lse, R, N, H, dqkv, delta)
e1.record(); torch.cuda.synchronize()
print(f"attention backward (R = {R}): {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call")
