"""GPU parity of the detector-training kernels against PyTorch fp32 autograd on the
CPU.  Parity UNPINNED against ultralytics (absent from the tree and the image): the reference here is the published
layer definition (Conv = conv -> BatchNorm2d(eps 1e-3, momentum 0.03) -> SiLU) executed by torch on bf16-representable
inputs; tolerances cover bf16 rounding of the outputs (2^-9 relative) and are written per test."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def yv():
    import yvhip
    yvhip.require_gpu()
    return yvhip


def bf(t):
    return t.to(torch.bfloat16)


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def nhwc(t):        # (B,C,H,W) -> (B,H,W,C) contiguous
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("B,H,C,off", [(2, 20, 64, 0), (3, 12, 48, 16), (1, 40, 16, 8), (2, 6, 256, 0)])
def test_bn_silu_forward_backward(yv, B, H, C, off):
    g = torch.Generator().manual_seed(B * 100 + H + C)
    T = B * H * H
    z = bf(torch.randn(B, C, H, H, generator=g) * 1.5 + 0.3).float()
    gamma = 1 + 0.2 * torch.randn(C, generator=g); beta = 0.2 * torch.randn(C, generator=g)
    res = bf(torch.randn(B, C, H, H, generator=g)).float()
    da = bf(torch.randn(B, C, H, H, generator=g)).float()
    rm0, rv0 = torch.randn(C, generator=g) * 0.1, 1 + 0.1 * torch.rand(C, generator=g)
    # reference
    zr = z.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    a_ref = F.silu(F.batch_norm(zr, rm, rv, gr, br, training=True, momentum=0.03, eps=1e-3)) + res
    a_ref.backward(da)
    # device: z lives at channel offset `off` of a wider buffer (views)
    ld = C + off + 8
    zb = torch.zeros(B, H, H, ld, dtype=torch.bfloat16, device=DEV); zb[..., off:off + C] = bf(nhwc(z)).to(DEV)
    ob = torch.zeros_like(zb); rb = torch.zeros_like(zb); rb[..., off:off + C] = bf(nhwc(res)).to(DEV)
    dab = torch.zeros_like(zb); dab[..., off:off + C] = bf(nhwc(da)).to(DEV)
    dzb = torch.zeros_like(zb)
    mean = torch.empty(C, device=DEV); rstd = torch.empty(C, device=DEV)
    rmd, rvd = rm0.to(DEV), rv0.to(DEV)
    ws = torch.empty(yv.bn_ws_floats(T, C), device=DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    yv.bn_stats(yv.mview(zb, off, C), T, mean, rstd, rmd, rvd, ws)
    yv.bn_act_fwd(yv.mview(zb, off, C), T, mean, rstd, gd, bd, yv.mview(ob, off, C), res=yv.mview(rb, off, C))
    dg = torch.empty(C, device=DEV); db = torch.empty(C, device=DEV)
    yv.bn_act_bwd(yv.mview(dab, off, C), yv.mview(zb, off, C), T, mean, rstd, gd, bd, dg, db, yv.mview(dzb, off, C), ws)
    torch.cuda.synchronize()
    mref = z.mean(dim=(0, 2, 3)); vref = z.var(dim=(0, 2, 3), unbiased=False)
    assert torch.allclose(mean.cpu(), mref, atol=2e-5, rtol=1e-5)
    assert torch.allclose(rstd.cpu(), 1 / torch.sqrt(vref + 1e-3), atol=1e-5, rtol=2e-5)
    assert torch.allclose(rmd.cpu(), rm, atol=1e-5, rtol=1e-5) and torch.allclose(rvd.cpu(), rv, atol=1e-5, rtol=2e-5)
    a_dev = nchw(ob[..., off:off + C].float().cpu())
    assert rel_l2(a_dev, a_ref.detach()) < 4e-3                       # two bf16 roundings (activation, sum)
    assert float(ob[..., :off].abs().sum()) == 0 and float(ob[..., off + C:].abs().sum()) == 0   # slice only
    assert rel_l2(nchw(dzb[..., off:off + C].float().cpu()), zr.grad) < 4e-3
    assert torch.allclose(dg.cpu(), gr.grad, atol=2e-3 * float(gr.grad.abs().max()), rtol=1e-3)
    assert torch.allclose(db.cpu(), br.grad, atol=2e-3 * float(br.grad.abs().max()), rtol=1e-3)
    # frozen statistics (eval-mode BatchNorm): no mean / variance terms in dz
    yv.bn_act_bwd(yv.mview(dab, off, C), yv.mview(zb, off, C), T, mean, rstd, gd, bd, dg, db, yv.mview(dzb, off, C), ws,
                  batch_stats=False)
    z2 = z.clone().requires_grad_(True)
    a2 = F.silu(F.batch_norm(z2, mref.clone(), vref.clone(), gamma, beta, training=False, eps=1e-3))
    a2.backward(da)
    assert rel_l2(nchw(dzb[..., off:off + C].float().cpu()), z2.grad) < 4e-3


def test_view_ops(yv):
    g = torch.Generator().manual_seed(5)
    B, H, W, C = 2, 6, 5, 24
    src = bf(torch.randn(B, H, W, C, generator=g)).to(DEV)
    big = bf(torch.randn(B, 2 * H, 2 * W, C + 8, generator=g)).to(DEV)
    # nearest 2x up into a channel slice
    dst = big.clone()
    yv.view_op(yv.VIEW_UP2, yv.mview(src), yv.mview(dst, 8, C), B, H, W)
    exp = big.clone(); exp[..., 8:] = src.repeat_interleave(2, 1).repeat_interleave(2, 2)
    assert torch.equal(dst, exp)
    # adjoint, accumulated
    acc0 = bf(torch.randn(B, H, W, C, generator=g)).to(DEV)
    acc = acc0.clone()
    yv.view_op(yv.VIEW_UP2_BWD, yv.mview(big, 8, C), yv.mview(acc), B, H, W)
    pooled = big[..., 8:].float().view(B, H, 2, W, 2, C)
    ref = acc0.float() + pooled[:, :, 0, :, 0] + pooled[:, :, 0, :, 1] + pooled[:, :, 1, :, 0] + pooled[:, :, 1, :, 1]
    assert torch.equal(acc, ref.to(torch.bfloat16))       # sequential f32 sum in the same order, one rounding
    # zero insertion
    zi = big.clone()
    yv.view_op(yv.VIEW_ZERO_INSERT, yv.mview(src), yv.mview(zi, 0, C), B, H, W)
    exp = big.clone(); exp[..., :C] = 0; exp[:, ::2, ::2, :C] = src
    assert torch.equal(zi, exp)
    # add / copy / zero
    d = acc0.clone(); yv.view_op(yv.VIEW_ADD, yv.mview(src), yv.mview(d), B, H, W)
    assert torch.equal(d, (acc0.float() + src.float()).to(torch.bfloat16))
    d = big.clone(); yv.view_op(yv.VIEW_ZERO, None, yv.mview(d, 8, 16), B, 2 * H, 2 * W)
    exp = big.clone(); exp[..., 8:24] = 0
    assert torch.equal(d, exp)
    d = torch.zeros_like(src); yv.view_op(yv.VIEW_COPY, yv.mview(src), yv.mview(d), B, H, W)
    assert torch.equal(d, src)


def test_maxpool5_backward_with_ties(yv):
    """Values from a 7-level grid: most windows hold several equal maxima; torch routes the gradient to the first."""
    g = torch.Generator().manual_seed(9)
    B, H, W, C = 2, 9, 11, 16
    x = (torch.randint(0, 7, (B, C, H, W), generator=g).float() * 0.5 - 1.0)
    dout = bf(torch.randn(B, C, H, W, generator=g)).float()
    xr = x.clone().requires_grad_(True)
    F.max_pool2d(xr, 5, 1, 2).backward(dout)
    din0 = bf(torch.randn(B, H, W, C, generator=g))
    din = din0.clone().to(DEV)
    xd, dd = bf(nhwc(x)).to(DEV), bf(nhwc(dout)).to(DEV)          # views hold raw pointers: keep the tensors alive
    yv.maxpool5_bwd(yv.mview(xd), yv.mview(dd), yv.mview(din), B, H, W)
    torch.cuda.synchronize()
    ref = din0.float() + nhwc(xr.grad)
    assert torch.allclose(din.float().cpu(), ref, atol=0.04, rtol=8e-3)       # up to 25 bf16 addends, one rounding
    assert rel_l2(din.float().cpu(), ref) < 3e-3


@pytest.mark.parametrize("B,Hin,Cin,Cout,k,s", [(2, 16, 16, 32, 3, 1), (2, 16, 32, 16, 3, 2), (1, 12, 64, 64, 1, 1),
                                                 (3, 10, 8, 24, 3, 2), (2, 20, 48, 8, 1, 1), (2, 8, 128, 128, 3, 1)])
def test_conv_backward(yv, B, Hin, Cin, Cout, k, s):
    """dgrad = conv of dz (zero-inserted for stride 2) with the flipped/transposed weight; wgrad = dz^T . im2col(x).
    Small-integer operands: every product and partial sum is exact, results must EQUAL torch autograd."""
    g = torch.Generator().manual_seed(B + Hin + Cin + Cout + k + s)
    Hout = (Hin - 1) // s + 1
    x = torch.randint(-2, 3, (B, Cin, Hin, Hin), generator=g).float()
    w = torch.randint(-2, 3, (Cout, Cin, k, k), generator=g).float()
    dz = torch.randint(-2, 3, (B, Cout, Hout, Hout), generator=g).float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, s, k // 2).backward(dz)
    taps = k * k
    wk = bf(w.permute(0, 2, 3, 1).reshape(Cout, taps * Cin)).contiguous().to(DEV)        # (Cout, taps, Cin)
    T = B * Hout * Hout
    Tp = (T + 63) // 64 * 64
    dzd = torch.zeros(Tp, Cout, dtype=torch.bfloat16, device=DEV); dzd[:T] = bf(nhwc(dz)).reshape(T, Cout).to(DEV)
    xd = bf(nhwc(x)).to(DEV)
    # ---- data gradient
    wd = torch.empty(Cin, taps * Cout, dtype=torch.bfloat16, device=DEV)
    yv.conv_weight_dgrad(wk, Cout, taps, Cin, wd)
    prev = bf(torch.randint(-3, 4, (B, Hin, Hin, Cin), generator=g).float()).to(DEV)     # gradient already in the slice
    dx = prev.clone()
    if s == 1:
        src = yv.yv_view(dzd.data_ptr(), Cout, Cout, 0)
    else:
        zi = torch.zeros(B, 2 * Hout, 2 * Hout, Cout, dtype=torch.bfloat16, device=DEV)
        yv.view_op(yv.VIEW_ZERO_INSERT, yv.mview(dzd), yv.mview(zi), B, Hout, Hout)
        assert 2 * Hout == Hin
        src = yv.mview(zi)
    yv.conv_view(src, B, Hin, Hin, k, 1, wd, Cin, yv.mview(dx), res=yv.mview(dx))          # dx = prev + dgrad
    assert torch.equal(dx.cpu(), bf(prev.float().cpu() + nhwc(xr.grad)))       # exact sum, one bf16 rounding
    # ---- weight gradient
    dw = torch.empty(Cout, taps * Cin, device=DEV)
    if k == 3:
        col = torch.zeros(Tp, 9 * Cin, dtype=torch.bfloat16, device=DEV)
        yv.im2col3(yv.mview(xd), B, Hin, Hin, s, col)
        ref_col = F.unfold(x, 3, padding=1, stride=s).view(B, Cin, 9, Hout * Hout).permute(0, 3, 2, 1).reshape(T, 9 * Cin)
        assert torch.equal(col[:T].float().cpu(), ref_col)
        yv.wgrad(dzd, col, dw, T=Tp)
    else:
        xp = torch.zeros(Tp, Cin, dtype=torch.bfloat16, device=DEV); xp[:T] = xd.reshape(T, Cin)
        yv.wgrad(dzd, xp, dw, T=Tp)
    assert torch.equal(dw.cpu(), wr.grad.permute(0, 2, 3, 1).reshape(Cout, taps * Cin))
    if k == 3 and s == 1:
        # ---- the same gradient without im2col: operands on the zero-padded pixel grid (yv_wgrad_conv3).  The margins and
        # the tail rows of the activation hold (finite) junk: they only ever meet zero rows of the padded dz
        hp = Hin + 2
        tpad = B * hp * hp
        tpp, mg = (tpad + 63) // 64 * 64, hp + 1
        buf = torch.randint(-3, 4, ((tpp + 2 * mg) * Cin,), generator=g).to(torch.bfloat16).to(DEV)
        xpad = buf[mg * Cin:(mg + tpp) * Cin].view(tpp, Cin)
        yv.view_op(yv.VIEW_PAD, yv.mview(xd), yv.mview(xpad), B, Hin, Hin)
        ref_pad = F.pad(nhwc(x), (0, 0, 1, 1, 1, 1)).reshape(tpad, Cin)
        assert torch.equal(xpad[:tpad].float().cpu(), ref_pad)
        dzp = torch.full((tpp, Cout), 7.0, dtype=torch.bfloat16, device=DEV)
        yv.view_op(yv.VIEW_PAD, yv.mview(dzd), yv.mview(dzp), B, Hout, Hout)
        dzp[tpad:].zero_()
        dw2 = torch.full((Cout, 9 * Cin), -1.0, device=DEV)
        yv.wgrad_conv3(dzp, xpad, dw2, tpp, hp)
        assert torch.equal(dw2.cpu(), wr.grad.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin))


def test_blob_nhwc8(yv):
    g = torch.Generator().manual_seed(2)
    img = torch.randint(0, 256, (2, 16, 16, 3), generator=g, dtype=torch.uint8)
    out = torch.empty(2, 16, 16, 8, dtype=torch.bfloat16, device=DEV)
    yv.blob_nhwc8(img.to(DEV), out)
    exp = torch.zeros(2, 16, 16, 8); exp[..., :3] = img.float() * torch.tensor(1.0 / 255.0)
    assert torch.equal(out.cpu(), exp.to(torch.bfloat16))


# ---------------------------------------------------------------------------------- whole network
def _oracle_state(scale, nc, seed):
    from oracle import yolo_train as oy
    sd = oy.init_train_state(scale, nc, seed)
    for k in list(sd):
        if k.endswith("conv.weight") or (k.endswith(".weight") and ".bn." not in k):
            sd[k] = bf(sd[k]).float()                      # both sides use the same (bf16-representable) conv weights
    return sd


def _act_nchw(a, t):
    return t[:a.T].float().cpu().view(a.B, a.H, a.W, a.C).permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("scale,nc,S,B", [("n", 5, 160, 2), ("s", 80, 640, 2)])
def test_trainer_local_consistency(yv, scale, nc, S, B):
    """Every module of the un-fused YOLOv8 (n: nc 5, 2 x 160 x 160; s: nc 80, 2 x 640 x 640 = the model, class count and
    resolution of BASELINE.json configs[3]) checked in isolation: the oracle module
    (oracle/yolo_train.py, fp32 math with bf16 storage where the device stores bf16) is run on the DEVICE's input
    activation and back-propagated from the DEVICE's output gradient; outputs, parameter gradients and (summed over
    all consumers of a tensor) input gradients must agree.  Random-init BatchNorm stacks amplify bf16 storage noise
    chaotically with depth (fp32 vs bf16 storage: ~10 % at the head, measured with the oracle alone), which is why the
    comparison is per module.  Tolerances: outputs rel-L2 <= 6e-3; gradients rel-L2 <= 3e-2 (the device also rounds
    every activation gradient to bf16, the oracle does not)."""
    from oracle import yolo_train as oy
    from oracle.yolo import topology
    from yvhip.yolo_training import YoloTrainer
    sd = _oracle_state(scale, nc, 3)
    g = torch.Generator().manual_seed(11)
    img = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8)
    tr = YoloTrainer({k: v.clone() for k, v in sd.items()}, scale=scale, nc=nc, size=S, batch=B)
    ncp = tr.ncp
    outs = tr.forward(img.to(DEV))
    R = []
    for s in range(3):
        T = outs[s][0].shape[0]
        db = bf(torch.randn(T, 64, generator=g)).float()
        dc = torch.zeros(T, ncp); dc[:, :nc] = bf(torch.randn(T, nc, generator=g)).float()
        R.append((db.to(DEV), dc.to(DEV)))
    tr.backward(R)
    torch.cuda.synchronize()
    got = tr.grads()
    P = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    contrib = {}
    out_err, grad_err = {}, {}

    def add(idx, t):
        contrib[idx] = contrib.get(idx, 0) + t

    x0 = _act_nchw(tr.x0, tr.x0.buf)[:, :3]
    topo = {i: (k, a) for i, k, a in topology(scale)}
    for idx, kind, p in tr.layers:
        m = tr.mod[idx]
        if idx == 0:
            xin = x0.clone().requires_grad_(True)
        elif "a" in p:                                               # neck: input = concat buffer (checked below)
            (ia, ua), (ib, _) = p["a"], p["b"]
            cat = tr.aux[idx]["cat"]
            xa = _act_nchw(tr.out[ia], tr.out[ia].buf)
            xb = _act_nchw(tr.out[ib], tr.out[ib].buf)
            exp = torch.cat([torch.nn.functional.interpolate(xa, scale_factor=2, mode="nearest") if ua else xa, xb], 1)
            assert torch.equal(_act_nchw(cat, cat.buf), exp), idx
            xin = exp.clone().requires_grad_(True)
        else:
            xin = _act_nchw(tr.out[idx - 1], tr.out[idx - 1].buf).requires_grad_(True)
        y = oy.run_module(P, idx, xin, scale, train=True, emulate_bf16=True)
        out_err[idx] = rel_l2(_act_nchw(tr.out[idx], tr.out[idx].buf), y.detach())
        y.backward(_act_nchw(tr.out[idx], tr.out[idx].grad))
        if idx == 0:
            pass
        elif "a" in p:
            ca = tr.out[ia].C
            ga = xin.grad[:, :ca]
            if ua:
                ga = ga.view(B, ca, ga.shape[2] // 2, 2, ga.shape[3] // 2, 2).sum((3, 5))
            add(ia, ga); add(ib, xin.grad[:, ca:])
        else:
            add(idx - 1, xin.grad)
    for s, fidx in enumerate((15, 18, 21)):
        f = _act_nchw(tr.out[fidx], tr.out[fidx].buf).requires_grad_(True)
        b, c = oy.run_detect_scale(P, s, f, train=True, emulate_bf16=True)
        h = b.shape[-1]
        out_err[f"det{s}.box"] = rel_l2(outs[s][0].cpu().view(B, h, h, 64).permute(0, 3, 1, 2), b.detach())
        out_err[f"det{s}.cls"] = rel_l2(outs[s][1].cpu().view(B, h, h, ncp).permute(0, 3, 1, 2)[:, :nc], c.detach())
        (b * R[s][0].cpu().view(B, h, h, 64).permute(0, 3, 1, 2)).sum().backward(retain_graph=True)
        (c * R[s][1].cpu().view(B, h, h, ncp).permute(0, 3, 1, 2)[:, :nc]).sum().backward()
        add(fidx, f.grad)
    for k, v in P.items():
        if v.grad is not None:
            grad_err[k] = rel_l2(got[k], v.grad)
    in_err = {idx: rel_l2(_act_nchw(tr.out[idx], tr.out[idx].grad), t) for idx, t in contrib.items()}
    worst_out = max(out_err.items(), key=lambda kv: kv[1])
    worst_grad = sorted(grad_err.items(), key=lambda kv: -kv[1])[:4]
    worst_in = max(in_err.items(), key=lambda kv: kv[1])
    assert worst_out[1] < 6e-3, worst_out
    assert worst_grad[0][1] < 3e-2, worst_grad
    assert worst_in[1] < 3e-2, worst_in
    assert len(grad_err) == 183 and len(in_err) == 16


def test_trainer_whole_network_direction(yv):
    """End to end against pure fp32 autograd with a linear probe loss: chaotic amplification of bf16 storage noise
    (see test_trainer_local_consistency) allows only a statistical statement - the gradient of every parameter tensor
    must point the same way (cosine >= 0.75, median >= 0.95)."""
    from oracle import yolo_train as oy
    from yvhip.yolo_training import YoloTrainer
    scale, nc, S, B = "n", 5, 160, 2
    sd = _oracle_state(scale, nc, 3)
    g = torch.Generator().manual_seed(11)
    img = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8)
    tr = YoloTrainer({k: v.clone() for k, v in sd.items()}, scale=scale, nc=nc, size=S, batch=B)
    outs = tr.forward(img.to(DEV))
    params = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    x = bf(img.float() * torch.tensor(1.0 / 255.0)).float().permute(0, 3, 1, 2).contiguous()
    ref = oy.forward_train(params, x, scale, nc, train=True)
    R, loss = [], 0.0
    for s, (rb, rc) in enumerate(ref):
        h = rb.shape[-1]
        rbx = bf(torch.randn(rb.shape, generator=g)).float(); rcl = bf(torch.randn(rc.shape, generator=g)).float()
        loss = loss + (rb * rbx).sum() + (rc * rcl).sum()
        dc = torch.zeros(B * h * h, 8)
        dc[:, :nc] = rcl.permute(0, 2, 3, 1).reshape(-1, nc)
        R.append((rbx.permute(0, 2, 3, 1).reshape(-1, 64).contiguous().to(DEV), dc.to(DEV)))
    loss.backward()
    tr.backward(R)
    torch.cuda.synchronize()
    got = tr.grads()
    cos = {k: float(torch.nn.functional.cosine_similarity(got[k].flatten().double(), v.grad.flatten().double(), dim=0))
           for k, v in params.items() if v.grad is not None and float(v.grad.abs().max()) > 0}
    vals = sorted(cos.values())
    assert vals[0] >= 0.75, sorted(cos.items(), key=lambda kv: kv[1])[:5]
    assert vals[len(vals) // 2] >= 0.95
    for k in ("model.0", "model.9.cv2", "model.22.cv3.2.1"):
        assert torch.allclose(tr.run_mean[k].cpu(), params[k + ".bn.running_mean"], atol=5e-3, rtol=5e-2)
        assert torch.allclose(tr.run_var[k].cpu(), params[k + ".bn.running_var"], atol=5e-3, rtol=5e-2)


@pytest.mark.parametrize("B,S,G,counts,seed", [(3, 160, 6, [4, 0, 6], 0), (2, 320, 12, [12, 7], 1), (2, 96, 3, [0, 0], 2),
                                              (4, 160, 5, [1, 5, 2, 3], 3)])
def test_detect_loss_vs_oracle(yv, B, S, G, counts, seed):
    """yv_detect_loss (assigner + CIoU/DFL/BCE + analytic gradient) against oracle/yolo_train.py::detection_loss and
    torch autograd on the same logits.  f32 on both sides: loss terms rel 2e-4, gradients rel-L2 2e-4.  Cases: images
    without boxes, a batch without any box (normaliser clamps to 1), many overlapping boxes (anchors claimed by several
    ground truths), boxes smaller than a stride-32 cell (no anchor inside)."""
    from oracle import yolo_train as oy
    nc, ncp = 5, 8
    g = torch.Generator().manual_seed(seed)
    hs = [S // 8, S // 16, S // 32]
    outs, dev_box, dev_cls = [], [], []
    for h in hs:
        b = (torch.randn(B, 64, h, h, generator=g) * 1.5).requires_grad_(True)
        c = (torch.randn(B, nc, h, h, generator=g) * 1.5 - 1.0).requires_grad_(True)
        outs.append((b, c))
        dev_box.append(b.detach().permute(0, 2, 3, 1).reshape(-1, 64).contiguous().to(DEV))
        cp = torch.zeros(B * h * h, ncp); cp[:, :nc] = c.detach().permute(0, 2, 3, 1).reshape(-1, nc)
        dev_cls.append(cp.to(DEV))
    ctr = torch.rand(B, G, 2, generator=g) * S
    wh = torch.rand(B, G, 2, generator=g) * S * 0.45 + 2.0
    wh[:, 0] = 6.0                                             # a box smaller than the coarse cells
    gtb = torch.cat([(ctr - wh / 2).clamp(0, S), (ctr + wh / 2).clamp(0, S)], -1)
    gtl = torch.randint(0, nc, (B, G), generator=g)
    cnt = torch.tensor(counts, dtype=torch.int32)
    mask = torch.arange(G)[None] < cnt[:, None]
    total, (lb, lc, ld) = oy.detection_loss(outs, gtl, gtb, mask, S, nc)
    total.backward()
    A = sum(h * h for h in hs)
    dbox = [torch.full_like(t, 7.0) for t in dev_box]; dcls = [torch.full_like(t, 7.0) for t in dev_cls]
    loss = torch.zeros(4, device=DEV)
    ws = torch.zeros(yv.detect_loss_ws_bytes(B, A, G), dtype=torch.uint8, device=DEV)
    yv.detect_loss(dev_box, dev_cls, dbox, dcls, B, S, nc, ncp, gtb.to(DEV), gtl.to(torch.int32).to(DEV), cnt.to(DEV), loss, ws)
    torch.cuda.synchronize()
    got = loss.cpu()
    for v, r in zip(got, (total.detach(), lb, lc, ld)):
        assert abs(float(v) - float(r)) <= 2e-4 * max(1.0, abs(float(r))), (got, total, lb, lc, ld)
    for s, h in enumerate(hs):
        gb = outs[s][0].grad if outs[s][0].grad is not None else torch.zeros_like(outs[s][0])   # no positives: unused
        rb = gb.permute(0, 2, 3, 1).reshape(-1, 64)
        rc = outs[s][1].grad.permute(0, 2, 3, 1).reshape(-1, nc)
        assert rel_l2(dbox[s].cpu(), rb) < 2e-4 or (float(rb.abs().max()) == 0 and float(dbox[s].abs().max()) == 0), s
        assert rel_l2(dcls[s].cpu()[:, :nc], rc) < 2e-4, s
        assert float(dcls[s][:, nc:].abs().max()) == 0


def test_training_steps_reduce_the_loss(yv):
    """End to end: 12 SGD steps of YoloTrainer.step on one fixed batch (YOLOv8n, nc 5, 4 x 160 x 160, 3 boxes per
    image) must bring the v8 loss down, and the state dict must round-trip through the un-fused ultralytics key layout."""
    from yvhip.yolo_training import YoloTrainer, init_yolo_train_state
    scale, nc, S, B, G = "n", 5, 160, 4, 3
    sd = init_yolo_train_state(scale, nc, seed=1)
    tr = YoloTrainer(sd, scale=scale, nc=nc, size=S, batch=B, lr=2e-3, momentum=0.9, weight_decay=5e-4)
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8).to(DEV)
    ctr = torch.rand(B, G, 2, generator=g) * (S - 60) + 30
    wh = torch.rand(B, G, 2, generator=g) * 50 + 20
    gtb = torch.cat([ctr - wh / 2, ctr + wh / 2], -1).to(DEV)
    gtl = torch.randint(0, nc, (B, G), generator=g, dtype=torch.int32).to(DEV)
    gtn = torch.full((B,), G, dtype=torch.int32, device=DEV)
    hist = []
    for _ in range(12):
        hist.append(tr.step(img, gtb, gtl, gtn).cpu().clone())
    assert all(torch.isfinite(h).all() for h in hist)
    first, last = float(hist[0][0]), float(hist[-1][0])
    assert last < 0.8 * first, [float(h[0]) for h in hist]
    out = tr.state_dict()
    assert set(out) == set(sd)
    tr2 = YoloTrainer(out, scale=scale, nc=nc, size=S, batch=B)
    assert torch.equal(tr2.P.cpu(), tr.P.cpu())


def test_trained_weights_fold_into_the_inference_engine(yv):
    """Train a few steps, save, load through YOLOTensorRT.models.fold_batchnorm into the inference YoloEngine (BN folded
    with the RUNNING statistics the trainer maintained) and compare its raw head outputs with the oracle's eval-mode
    forward on the same state dict: closes the loop utils.trainYolo.train -> app.py inference.  Tolerance rel-L2 5e-2
    (bf16 activations through 23 modules; eval-mode BN does not amplify like batch statistics do)."""
    from oracle import yolo_train as oy
    from YOLOTensorRT.models import fold_batchnorm
    from yvhip import engines
    from yvhip.yolo_training import YoloTrainer, init_yolo_train_state
    scale, nc, S, B, G = "n", 5, 160, 4, 2
    tr = YoloTrainer(init_yolo_train_state(scale, nc, seed=2), scale=scale, nc=nc, size=S, batch=B, lr=1e-3)
    g = torch.Generator().manual_seed(8)
    img = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8)
    ctr = torch.rand(B, G, 2, generator=g) * 100 + 30
    gtb = torch.cat([ctr - 20, ctr + 20], -1).to(DEV)
    gtl = torch.randint(0, nc, (B, G), generator=g, dtype=torch.int32).to(DEV)
    gtn = torch.full((B,), G, dtype=torch.int32, device=DEV)
    for _ in range(40):                                   # momentum 0.03: running statistics need a few dozen updates
        tr.step(img.to(DEV), gtb, gtl, gtn)
    sd = tr.state_dict()
    eng = engines.YoloEngine(fold_batchnorm(sd), scale, nc, S, device=DEV)
    box_l, cls_l = eng.forward_raw(img.to(DEV))
    torch.cuda.synchronize()
    x = bf(img.float() * torch.tensor(1.0 / 255.0)).float().permute(0, 3, 1, 2).contiguous()
    ref = oy.forward_train({k: v.clone() for k, v in sd.items()}, x, scale, nc, train=False)
    for s, (rb, rc) in enumerate(ref):
        assert rel_l2(box_l[s].float().cpu().permute(0, 3, 1, 2), rb) < 5e-2, s
        assert rel_l2(cls_l[s].float().cpu().permute(0, 3, 1, 2)[:, :nc], rc) < 5e-2, s


@pytest.mark.parametrize("kind", ["sgd_nesterov", "adamw"])
def test_optimisers_match_torch(yv, kind):
    """yv_optim_step against torch.optim.SGD(nesterov=True) / torch.optim.AdamW on the CPU (the optimisers ultralytics
    instantiates), 4 steps with changing lr; yv_ema_update against ModelEMA's rule.  fp32, same operation order:
    max relative difference 1e-5 (torch's vectorised CPU kernels contract `p + alpha*g` into an FMA, the HIP build keeps
    the two roundings: measured 3e-6)."""
    g = torch.Generator().manual_seed(4)
    n = 10007
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * (0.5 + i) for i in range(4)]
    lrs = [1e-3, 2e-3, 5e-4, 1e-3]
    ref = torch.nn.Parameter(p0.clone())
    if kind == "adamw":
        opt = torch.optim.AdamW([ref], lr=lrs[0], betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
        code, b1, wd = yv.OPT_ADAMW, 0.9, 5e-4
    else:
        opt = torch.optim.SGD([ref], lr=lrs[0], momentum=0.937, nesterov=True, weight_decay=5e-4)
        code, b1, wd = yv.OPT_SGD_NESTEROV, 0.937, 5e-4
    p = p0.clone().to(DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    mirror = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    ema_ref, ema = p0.clone(), p0.clone().to(DEV)
    for t, (gr, lr) in enumerate(zip(grads, lrs), start=1):
        for grp in opt.param_groups:
            grp["lr"] = lr
        ref.grad = gr.clone()
        opt.step()
        yv.optim_step(code, p, gr.to(DEV), m, v, lr, t, beta1=b1, weight_decay=wd, mirror=mirror)
        d = 0.9999 * (1 - math.exp(-t / 2000))
        ema_ref.mul_(d).add_((1 - d) * ref.detach())
        yv.ema_update(ema, p, d)
    torch.cuda.synchronize()
    err = float(((p.cpu() - ref.detach()).abs() / (ref.detach().abs() + 1e-3)).max())
    assert err < 1e-5, err
    assert torch.equal(mirror.cpu(), p.cpu().to(torch.bfloat16))
    assert torch.allclose(ema.cpu(), ema_ref, rtol=1e-6, atol=1e-7)


def test_validate_metrics_on_planted_detections(yv, tmp_path):
    """yvhip.yolo_val.validate end to end with a detector whose head is rigged to fire exactly on the ground truth is
    not constructible from random weights; instead check the pipeline pieces it composes on real kernels: the engine +
    NMS outputs for a trained-for-a-while model are finite, sorted by score, inside the letterboxed image, and the
    metric dictionary is consistent (0 <= mAP50-95 <= mAP50 <= 0.995; detections counted; no ground truth -> zeros)."""
    from PIL import Image
    import numpy as np
    from yvhip.yolo_training import init_yolo_train_state
    from yvhip.yolo_val import validate
    rng = np.random.default_rng(3)
    root = tmp_path / "d"
    (root / "images" / "val").mkdir(parents=True); (root / "labels" / "val").mkdir(parents=True)
    samples = []
    for i in range(5):
        Image.fromarray(rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)).save(root / "images" / "val" / f"v{i}.png")
        (root / "labels" / "val" / f"v{i}.txt").write_text(f"{i % 5} 0.5 0.5 0.4 0.5\n")
        samples.append((str(root / "images" / "val" / f"v{i}.png"), str(root / "labels" / "val" / f"v{i}.txt")))
    sd = init_yolo_train_state("n", 5, seed=1)
    for k in sd:
        if ".cv3." in k and k.endswith(".2.bias"):
            sd[k] = sd[k] + 6.0                         # random head but confident: plenty of candidates above conf
    res = validate(sd, samples, "n", 5, size=128, batch=2, conf=0.25, iou=0.6)
    assert res["images"] == 5 and res["instances"] == 5 and res["detections"] > 0
    assert 0.0 <= res["map50_95"] <= res["map50"] <= 0.995 and 0.0 <= res["precision"] <= 1.0 and 0.0 <= res["recall"] <= 1.0
    assert sorted(res["ap_per_class"]) == [0, 1, 2, 3, 4]
    none = validate(sd, [], "n", 5, size=128)
    assert none["images"] == 0 and none["map50"] == 0.0
