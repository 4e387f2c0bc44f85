"""Diagnostic (not a test): where do a half-batch run and the union-batch run part ways?  Single process, no process
group: trainer A takes samples [0, R/2), trainer B samples [R/2, R), their gradients are summed by hand (what the
all-reduce does) and applied with grad_scale 1/2; trainer U takes the union batch.  Prints, per step, the worst
per-tensor difference of logits, gradients and parameters, plus run-to-run reproducibility of U."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import torch  # noqa: E402
from dp_rehearsal_worker import make_batch  # noqa: E402
from yvhip import sgd_step  # noqa: E402
from yvhip.training import VitTrainer  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def worst(da, db):
    e = {k: rel(da[k], db[k]) for k in da if float(db[k].abs().max()) > 0}
    k = max(e, key=e.get)
    return f"{e[k]:.1e} ({k})"


name, R, steps = sys.argv[1] if len(sys.argv) > 1 else "vit_base_patch16_224", 4, 3
sd, patches, labels, tok = make_batch(name, R)
dev = "cuda:0"
pm, lb = patches.to(dev), labels.to(dev)
h = R // 2
U, U2 = VitTrainer(sd, name, 5), VitTrainer(sd, name, 5)
A, B = VitTrainer(sd, name, 5), VitTrainer(sd, name, 5)
for s in range(steps):
    lu = U.forward(pm, R).clone(); U.backward(pm, lb, R)
    lu2 = U2.forward(pm, R).clone(); U2.backward(pm, lb, R)
    la = A.forward(pm[:h * tok], h).clone(); A.backward(pm[:h * tok], lb[:h], h)
    lbb = B.forward(pm[h * tok:], h).clone(); B.backward(pm[h * tok:], lb[h:], h)
    torch.cuda.synchronize()
    print(f"step {s}: logits halves vs union {rel(torch.cat([la, lbb]), lu):.1e}; union run-to-run {rel(lu2, lu):.1e}")
    gu, gu2 = U.grad_dict(), U2.grad_dict()
    gsum = {k: (A.grad_dict()[k] + B.grad_dict()[k]) * 0.5 for k in gu}
    print(f"        gradient: mean of halves vs union {worst(gsum, gu)}; union run-to-run {worst(gu2, gu)}")
    # apply: A and B both take the summed gradient with grad_scale 1/2 (what the all-reduce + SGD kernel do)
    tot = A.G + B.G
    A.G.copy_(tot); B.G.copy_(tot)
    for T, gs in ((U, 1.0), (U2, 1.0), (A, 0.5), (B, 0.5)):
        sgd_step(T.P, T.G, T.Mo, 0.01, T.momentum, T.wd, first=T.steps == 0, grad_scale=gs, mirror=T.P16)
        T.steps += 1
        T.refresh_working_copies()
    torch.cuda.synchronize()
    print(f"        params after the step: halves vs union {worst(A.state_dict(), U.state_dict())}; A vs B "
          f"{worst(A.state_dict(), B.state_dict())}; union run-to-run {worst(U2.state_dict(), U.state_dict())}")
