"""Diagnostic: ViT-B/16 trainer forward/backward against fp32 autograd with the shipped attention form and with the
round-1 single-tile form (yv_attention_debug(4)); prints logits error and the gradient error table head."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd")); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import yvhip
from oracle import boxes as ob, train as ot, vit as ov
from yvhip.training import VitTrainer

def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
name, R = "vit_base_patch16_224", 4
sd = ov.init_wrapper_state(name, seed=21)
g = torch.Generator().manual_seed(R)
x = (torch.rand(R, 3, 224, 224, generator=g) * 2 - 1).to(torch.bfloat16).float()
labels = torch.randint(0, 5, (R,), generator=g, dtype=torch.int32)
p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
logits = ov.wrapper_forward(p, x, name)
ot.build_loss(logits, F.one_hot(labels.long(), 5).float()).backward()
pm = torch.cat([torch.from_numpy(ob.patchify(x[r].numpy(), 16)) for r in range(R)]).to(torch.bfloat16).to("cuda:0")
for abl in (0, 4, 0):
    yvhip.lib.yv_attention_debug(abl)
    tr = VitTrainer(sd, name, 5)
    lg = tr.forward(pm, R).clone()
    tr.backward(pm, labels.to("cuda:0"), R)
    torch.cuda.synchronize()
    got = tr.grad_dict()
    err = sorted(((rel(got[k].cpu(), v.grad), k) for k, v in p.items()), reverse=True)
    vals = sorted(e for e, _ in err)
    print(f"attention form {abl}: logits rel-L2 {rel(lg.cpu(), logits.detach()):.4f}; gradient median {vals[len(vals)//2]:.4f} max {err[0][0]:.4f} ({err[0][1]}); "
          f"last-layer fc.3.weight {rel(got['fc.3.weight'].cpu(), p['fc.3.weight'].grad):.4f} blocks.11.mlp.fc2.weight {rel(got['model.blocks.11.mlp.fc2.weight'].cpu(), p['model.blocks.11.mlp.fc2.weight'].grad):.4f}")
yvhip.lib.yv_attention_debug(0)
