"""Weight-gradient GEMM (yv_wgrad: gemm_tn_kernel + splitk_reduce_kernel) at the ViT-B/16 fine-tune shapes (32 images = 6,304 tokens,
padded to 6,336), by number of token slices ("wgrad_split"; 0 = the shipped heuristic)."""
import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "yolov8-vit_amd"))
import torch
import yvhip
dev = "cuda:0"
T = int(os.environ.get("WG_T", 6336))
g = torch.Generator().manual_seed(0)
shapes = [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]
splits = [int(v) for v in os.environ.get("WG_SPLITS", "0,2,3,4,5,6,9").split(",")]
flush = torch.zeros(64 * 1024 * 1024, device=dev)
tot = {s: 0.0 for s in splits}
for name, N, K in shapes:
    dy = (torch.randn(T, N, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    x = torch.randn(T, K, generator=g).to(torch.bfloat16).to(dev)
    dw = torch.zeros(N, K, device=dev)
    ref = None
    for S in splits:
        yvhip.set_option("wgrad_split", S)
        yvhip.wgrad(dy, x, dw); torch.cuda.synchronize()
        if ref is None:
            ref = dw.clone()
            exact = dy.float().t() @ x.float()
            err = float((ref - exact).abs().max() / exact.abs().max())
        ts = []
        for rd in range(5):
            flush.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                yvhip.wgrad(dy, x, dw)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        t = sorted(ts)[2]
        tot[S] += t
        print(f"{name:5s} dW {N}x{K}  S={S}: {t:7.1f} us  {2.0 * T * N * K / t * 1e-6:7.1f} TF/s   max |d| vs S=first {float((dw - ref).abs().max()):.3g} (rel err of first vs fp32 {err:.2g})")
yvhip.set_option("wgrad_split", 0)
print("layer sums (us): " + "  ".join(f"S={s}: {v:.1f}" for s, v in tot.items()))
