import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "yolov8-vit_amd")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from oracle import boxes as ob, vit as ov, train as ot
from yvhip.training import VitTrainer
name, R = "vit_tiny_test", 8
sd = ov.init_wrapper_state(name, seed=21)
g = torch.Generator().manual_seed(R)
x = (torch.rand(R, 3, 224, 224, generator=g) * 2 - 1).to(torch.bfloat16).float()
labels = torch.randint(0, 5, (R,), generator=g, dtype=torch.int32)
p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
logits = ov.wrapper_forward(p, x, name)
loss = ot.build_loss(logits, F.one_hot(labels.long(), 5).float()); loss.backward()
tr = VitTrainer(sd, name, 5)
pm = torch.cat([torch.from_numpy(ob.patchify(x[r].numpy(), tr.P_)) for r in range(R)]).to(torch.bfloat16).cuda()
tr.forward(pm, R); tr.backward(pm, labels.cuda(), R); torch.cuda.synchronize()
got = tr.grad_dict()
for k, v in p.items():
    e = float((got[k].cpu().double() - v.grad.double()).norm() / (v.grad.double().norm() + 1e-30))
    print(f"{k:45s} {e:.4f}  |g|={float(v.grad.norm()):.3e}")
