import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/yolov8-vit_amd")
import torch
from oracle import yolo_train as oy
from yvhip.yolo_training import YoloTrainer
bf = lambda t: t.to(torch.bfloat16)
def rel(a, b): return float((a.double()-b.double()).norm()/(b.double().norm()+1e-30))
scale, nc, S, B = "n", 5, int(os.environ.get("DBG_S", 160)), int(os.environ.get("DBG_B", 2))
sd = oy.init_train_state(scale, nc, 3)
for k in list(sd):
    if k.endswith("conv.weight") or (k.endswith(".weight") and ".bn." not in k):
        sd[k] = bf(sd[k]).float()
g = torch.Generator().manual_seed(11)
img = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8)
tr = YoloTrainer({k: v.clone() for k, v in sd.items()}, scale=scale, nc=nc, size=S, batch=B)
outs = tr.forward(img.to("cuda:0")); torch.cuda.synchronize()
x = bf(img.float() * torch.tensor(1.0 / 255.0)).float().permute(0, 3, 1, 2).contiguous()
params = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
ref, feats = oy.forward_train(params, x, scale, nc, train=True, return_feats=True, emulate_bf16=os.environ.get('DBG_EMU', '1') == '1')
for idx in sorted(tr.out):
    a = tr.out[idx]
    dev = a.buf[:a.T].float().cpu().view(B, a.H, a.W, a.C).permute(0, 3, 1, 2)
    print(idx, tuple(dev.shape), "rel", round(rel(dev, feats[idx].detach()), 4))

R, loss = [], 0.0
for s_, (rb, rc) in enumerate(ref):
    h = rb.shape[-1]
    box_dev = outs[s_][0].cpu().view(B, h, h, 64).permute(0, 3, 1, 2)
    cls_dev = outs[s_][1].cpu().view(B, h, h, 8).permute(0, 3, 1, 2)
    print("scale", s_, "box rel", round(rel(box_dev, rb.detach()), 4), "cls rel", round(rel(cls_dev[:, :nc], rc.detach()), 4))
    rbx = bf(torch.randn(rb.shape, generator=g)).float(); rcl = bf(torch.randn(rc.shape, generator=g)).float()
    loss = loss + (rb * rbx).sum() + (rc * rcl).sum()
    db = rbx.permute(0, 2, 3, 1).reshape(-1, 64).contiguous().to("cuda:0")
    dc = torch.zeros(B * h * h, 8); dc[:, :nc] = rcl.permute(0, 2, 3, 1).reshape(-1, nc)
    R.append((db, dc.to("cuda:0")))
loss.backward()
tr.backward(R); torch.cuda.synchronize()
got = tr.grads()
errs = {k: rel(got[k], v.grad) for k, v in params.items() if v.grad is not None}
for k, e in sorted(errs.items(), key=lambda kv: -kv[1])[:12]: print("grad", k, round(e, 4))
vals = sorted(errs.values()); print("median", round(vals[len(vals)//2], 4), "n", len(vals))
for k in ["model.0.conv.weight", "model.2.m.0.cv1.conv.weight", "model.9.cv1.conv.weight", "model.12.cv1.conv.weight", "model.22.cv2.0.2.weight", "model.22.cv3.2.2.bias", "model.22.cv2.1.0.bn.weight"]:
    print("grad", k, round(errs[k], 4))
