"""Diagnostic: attention forward (output, log-sum-exp) and backward errors of the two single-tile forms against fp64."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import math, torch
import yvhip
dev = "cuda:0"
def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
R, N, H = 4, 197, 12
D = H * 64
g = torch.Generator().manual_seed(1)
for scale_in in (1.0, 3.0):
    qkv = (torch.randn(R * N, 3 * D, generator=g) * scale_in).to(torch.bfloat16)
    do = torch.randn(R * N, D, generator=g).to(torch.bfloat16)
    t = qkv.double().clone().requires_grad_(True)
    tt = t.view(R, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    sc = (tt[0] * 0.125) @ tt[1].transpose(-2, -1)
    att = sc.softmax(-1)
    ref_o = (att @ tt[2]).transpose(1, 2).reshape(R * N, D)
    ref_o.backward(do.double())
    ref_lse = torch.logsumexp(sc, -1) / math.log(2.0)          # (R,H,N), log2 domain
    for abl in (0, 4):
        yvhip.lib.yv_attention_debug(abl)
        out = torch.zeros(R * N, D, dtype=torch.bfloat16, device=dev); lse = torch.zeros(R * H * N, device=dev)
        yvhip.attention_train(qkv.to(dev), R, N, H, out, lse)
        dqkv = torch.zeros(R * N, 3 * D, dtype=torch.bfloat16, device=dev); dws = torch.zeros(R * H * N, device=dev)
        yvhip.attention_bwd(qkv.to(dev), out, do.to(dev), lse, R, N, H, dqkv, dws)
        torch.cuda.synchronize()
        le = (lse.cpu().double().view(R, H, N) - ref_lse.detach()).abs()
        got = dqkv.cpu().double()
        print(f"input scale {scale_in} form {abl}: out rel-L2 {rel(out.cpu(), ref_o.detach()):.2e}; lse abs err max {float(le.max()):.2e} mean {float(le.mean()):.2e}; "
              f"dq {rel(got[:, :D], t.grad[:, :D]):.2e} dk {rel(got[:, D:2*D], t.grad[:, D:2*D]):.2e} dv {rel(got[:, 2*D:], t.grad[:, 2*D:]):.2e}")
yvhip.lib.yv_attention_debug(0)
