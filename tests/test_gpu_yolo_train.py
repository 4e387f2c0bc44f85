"""GPU parity of the detector-training kernels against PyTorch fp32 autograd on the
CPU.  Parity UNPINNED against ultralytics (absent from the tree and the image): the reference here is the published
layer definition (Conv = conv -> BatchNorm2d(eps 1e-3, momentum 0.03) -> SiLU) executed by torch on bf16-representable
inputs; tolerances cover bf16 rounding of the outputs (2^-9 relative) and are written per test."""
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


def test_blob_nhwc8(yv):
    g = torch.Generator().manual_seed(2)
    img = torch.randint(0, 256, (2, 16, 16, 3), generator=g, dtype=torch.uint8)
    out = torch.empty(2, 16, 16, 8, dtype=torch.bfloat16, device=DEV)
    yv.blob_nhwc8(img.to(DEV), out)
    exp = torch.zeros(2, 16, 16, 8); exp[..., :3] = img.float() * torch.tensor(1.0 / 255.0)
    assert torch.equal(out.cpu(), exp.to(torch.bfloat16))
