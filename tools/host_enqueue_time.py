"""Host time to ENQUEUE one training step (no synchronisation inside the timed region) against the step's GPU time: is the
Python side ahead of the GPU?  MODE=yolo (YOLOv8s, 16 x 640 x 640) or vit (ViT-B/16 fine-tune, 32 crops)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "yolov8-vit_amd"))
import torch
dev = "cuda:0"
mode = os.environ.get("MODE", "yolo")
if mode == "yolo":
    from yvhip.yolo_training import YoloTrainer, init_yolo_train_state
    B, S, nc, G = 16, 640, 80, 8
    tr = YoloTrainer(init_yolo_train_state("s", nc, seed=42), scale="s", nc=nc, size=S, batch=B, lr=1e-4, device=dev)
    g = torch.Generator().manual_seed(4321)
    images = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8).to(dev)
    ctr = torch.rand(B, G, 2, generator=g) * S
    wh = torch.rand(B, G, 2, generator=g) * 240 + 16
    gtb = torch.cat([(ctr - wh / 2).clamp(0, S), (ctr + wh / 2).clamp(0, S)], -1).to(dev)
    gtl = torch.randint(0, nc, (B, G), generator=g, dtype=torch.int32).to(dev)
    gtn = torch.full((B,), G, dtype=torch.int32).to(dev)
    step = lambda: tr.step(images, gtb, gtl, gtn)
else:
    from yvhip import engines
    from yvhip.training import VitTrainer
    name, R = "vit_base_patch16_224", 32
    tr = VitTrainer(engines.init_vit_wrapper_state(name, 5, seed=42), name, 5, device=dev)
    g = torch.Generator().manual_seed(1)
    patches = (torch.rand(R * 196, 768, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    labels = torch.randint(0, 5, (R,), generator=g, dtype=torch.int32).to(dev)
    step = lambda: tr.step(patches, labels, 1e-4)
for _ in range(3):
    step()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
back = (time.perf_counter() - t0) / 10 * 1e3
print(f"{mode}: host enqueue of one step {sorted(enq)[4]:.2f} ms; that step start -> GPU idle {sorted(tot)[4]:.2f} ms; 10 steps back to back {back:.2f} ms per step")
