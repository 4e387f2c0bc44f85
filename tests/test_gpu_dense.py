"""GPU parity of the dense kernels (MFMA GEMM / implicit-GEMM conv / attention / LayerNorm /
head / pool / stem) against plain PyTorch fp32 on the CPU.  Inputs are bf16-representable so
the only differences are accumulation order and the bf16 rounding of the OUTPUT:
tolerances are written per test."""
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


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 768, 768), (197 * 3, 2304, 768), (130, 1000 + 24, 768),
                                   (64, 16, 144), (257, 40, 72), (1000, 64, 576)])
def test_linear_plain(yv, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a = bf(torch.randn(M, K, generator=g)); w = bf(torch.randn(N, K, generator=g) * 0.05)
    bias = torch.randn(N, generator=g)
    ref = a.float() @ w.float().t() + bias
    out = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    yv.linear(a.to(DEV), w.to(DEV), bias.to(DEV), out, flags=yv.EPI_OUT_F32)
    # f32 accumulate of exact bf16 products: only summation order differs
    assert torch.allclose(out.cpu(), ref, atol=1e-3, rtol=1e-4)
    outb = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    yv.linear(a.to(DEV), w.to(DEV), bias.to(DEV), outb)
    assert rel_l2(outb.cpu().float(), ref) < 3e-3                 # bf16 output rounding (2^-9)


def test_linear_epilogues(yv):
    g = torch.Generator().manual_seed(3)
    M, N, K = 394, 256, 128
    a = bf(torch.randn(M, K, generator=g)); w = bf(torch.randn(N, K, generator=g) * 0.1)
    bias = torch.randn(N, generator=g)
    lin = a.float() @ w.float().t() + bias
    # exact-erf GELU -> bf16
    o = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    yv.linear(a.to(DEV), w.to(DEV), bias.to(DEV), o, flags=yv.EPI_GELU)
    assert rel_l2(o.cpu().float(), F.gelu(lin)) < 4e-3
    # f32 residual stream read-modify-write
    x = torch.randn(M, N, generator=g)
    xd = x.to(DEV)
    yv.linear(a.to(DEV), w.to(DEV), bias.to(DEV), xd, flags=yv.EPI_RES_F32)
    assert torch.allclose(xd.cpu(), x + lin, atol=1e-3, rtol=1e-4)
    # patch-embed row remap + pos_embed (tok = 197-1 -> small: tok = 2)
    tok = 2
    pos = torch.randn(tok + 1, N, generator=g)
    R = M // tok
    xo = torch.full((R * (tok + 1), N), 7.0, device=DEV)
    yv.linear(a.to(DEV), w.to(DEV), bias.to(DEV), xo, flags=yv.EPI_OUT_F32 | yv.EPI_POSEMB, pos=pos.to(DEV), tok=tok)
    exp = torch.full((R, tok + 1, N), 7.0)
    exp[:, 1:] = lin.view(R, tok, N) + pos[1:]
    assert torch.allclose(xo.cpu().view(R, tok + 1, N), exp, atol=1e-3, rtol=1e-4)
    # device-side dynamic M
    md = torch.tensor([3], dtype=torch.int32, device=DEV)
    o2 = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    yv.linear(a.to(DEV), w.to(DEV), bias.to(DEV), o2, flags=yv.EPI_OUT_F32, m_dev=md, m_mul=50)
    assert torch.allclose(o2.cpu()[:150], lin[:150], atol=1e-3, rtol=1e-4) and float(o2[150:].abs().sum()) == 0


@pytest.mark.parametrize("M,N,K,kind", [
    (25216 // 4, 768, 768, "res"), (6304, 2304, 768, "plain"), (6304, 3072, 768, "gelu"), (6304, 768, 3072, "res"),
    (1000, 768, 768, "res"), (1000, 1000, 768, "gelu"), (130, 256, 1024, "plain"), (128 * 70 + 5, 640, 768, "gelu"),
    (70000, 384, 768, "plain")])
def test_linear_exact_integer(yv, M, N, K, kind):
    """Small-integer operands make every f32 sum exact, so the forward linears (bias -> bf16, bias + GELU -> bf16,
    bias + f32 residual read-modify-write) must EQUAL the integer reference element for element: any indexing,
    swizzle or edge-tile error shows as a wrong integer.  Shapes: many tiles, ragged M / N edges, device-side M."""
    g = torch.Generator().manual_seed(M * 7 + N + K)
    a = torch.randint(-2, 3, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    bias = torch.randint(-8, 9, (N,), generator=g).float()
    lin = a @ w.t() + bias                                     # exact: |sum| < 2^24
    ad, wd, bd = bf(a).to(DEV), bf(w).to(DEV), bias.to(DEV)
    flags = {"plain": 0, "gelu": yv.EPI_GELU, "res": yv.EPI_RES_F32}[kind]
    x = torch.randint(-64, 65, (M, N), generator=g).float()

    def run(m_dev=None, m_mul=1):
        o = x.clone().to(DEV) if kind == "res" else torch.full((M, N), 3.0, dtype=torch.bfloat16, device=DEV)
        yv.linear(ad, wd, bd, o, flags=flags, m_dev=m_dev, m_mul=m_mul)
        torch.cuda.synchronize()
        return o.cpu().float()

    got = run()
    if kind == "res":
        assert torch.equal(got, x + lin)
    elif kind == "plain":
        assert torch.equal(got, bf(lin).float())
    else:                                                      # exact argument, erf-GELU within bf16 rounding
        assert rel_l2(got, F.gelu(lin)) < 3e-3
    # device-side row count: rows past it keep their old contents
    md = torch.tensor([M // 3], dtype=torch.int32, device=DEV)
    part = run(md, 2)
    cut = 2 * (M // 3)
    assert torch.equal(part[:cut], got[:cut])
    assert torch.equal(part[cut:], x[cut:] if kind == "res" else torch.full((M - cut, N), 3.0))


@pytest.mark.parametrize("M,N,K,kind,rows", [
    (6304, 2304, 768, "plain", 0), (6304, 3072, 768, "gelu", 0), (25216, 2304, 768, "plain", 0), (25216, 3072, 768, "gelu", 256),
    (2048 + 37, 1536, 128, "plain", 256), (70000, 256, 192, "plain", 0), (12608, 4096, 1024, "gelu", 0), (300, 512, 64 * 5, "plain", 256),
    (25216, 768, 768, "res", 0), (25216, 768, 3072, "res", 0), (12608, 768, 768, "res", 0), (6304, 2304, 768, "plain", 128),
    (6304, 768, 768, "res", 160), (6304, 2304, 768, "plain", 192), (6304, 3072, 768, "gelu", 224), (12608, 768, 3072, "res", 224),
    (5000, 256, 128, "f32", 160), (9999, 512, 192, "res", 128),
    (6304, 768, 768, "res", 96), (6304, 768, 3072, "plain", 96), (9999, 512, 192, "res", 96), (5000, 256, 128, "f32", 96),
    (300, 512, 64 * 5, "plain", 128), (70000, 256, 192, "gelu", 96), (6304, 768, 2304, "plain", 0)])
@pytest.mark.parametrize("variant", [9, 11])
def test_linear_persistent_8phase_exact_integer(yv, M, N, K, kind, rows, variant):
    """variant 11 = gemm_p9_kernel (round 3: the free-running form - one barrier per K tile, stores straight from the accumulators
    with the weight rows permuted on the DMA source side; tile heights 96 .. 256 - the 96 / 128-row forms have two phases per K tile
    and fetch a whole K tile in one of them); variant 9 =
    gemm_p8_kernel (persistent 8-phase kernel: 128..256 x 256 tiles, LDS-DMA stream running across tile boundaries) forced
    through yv_set_option("linear_variant", 9); `rows` forces the tile height (0 = the host's choice).  Exact integer operands:
    every wrong offset, stage parity (odd K-tile counts: K = 192, 320), cross-tile prefetch into the wrong tile, ragged last M
    tile (zeros through the buffer range check), dummy DMA slot of the narrow tiles or slab mix-up is a wrong integer.  Shapes
    cover 1 .. 5 tiles per workgroup, the minimum K (two K tiles), N = 256 (one tile column), every epilogue (bias -> bf16,
    bias + GELU -> bf16, bias + f32 residual read-modify-write, plain f32) and the device-side row count."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    a = torch.randint(-2, 3, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    bias = torch.randint(-8, 9, (N,), generator=g).float()
    lin = a @ w.t() + bias
    ad, wd, bd = bf(a).to(DEV), bf(w).to(DEV), bias.to(DEV)
    flags = {"plain": 0, "gelu": yv.EPI_GELU, "res": yv.EPI_RES_F32, "f32": yv.EPI_OUT_F32}[kind]
    f32 = kind in ("res", "f32")
    x = torch.randint(-64, 65, (M, N), generator=g).float()
    yv.set_option("linear_variant", variant)
    yv.set_option("linear_p8_rows", rows)
    try:
        def run(m_dev=None, m_mul=1, with_bias=True):
            o = x.clone().to(DEV) if f32 else torch.full((M, N), 3.0, dtype=torch.bfloat16, device=DEV)
            yv.linear(ad, wd, bd if with_bias else None, o, flags=flags, m_dev=m_dev, m_mul=m_mul)
            torch.cuda.synchronize()
            return o.cpu().float()
        got = run()
        if kind == "plain":
            assert torch.equal(got, bf(lin).float())
            assert torch.equal(run(with_bias=False), bf(lin - bias).float())
        elif kind == "res":
            assert torch.equal(got, x + lin)
        elif kind == "f32":
            assert torch.equal(got, lin)
        else:
            # sigmoid-form GELU (max abs error 2.5e-5 against the erf form) + bf16 rounding of the output
            ref = F.gelu(lin)
            assert rel_l2(got, ref) < 3e-3
            assert float((got - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) + 1e-4
        md = torch.tensor([M // 3], dtype=torch.int32, device=DEV)
        part = run(md, 2)
        cut = 2 * (M // 3)
        assert torch.equal(part[:cut], got[:cut])
        assert torch.equal(part[cut:], x[cut:] if f32 else torch.full((M - cut, N), 3.0))
        # the 128 x 128 kernel must agree on the same data (bit for bit except for the GELU form)
        yv.set_option("linear_variant", 1)
        o = x.clone().to(DEV) if f32 else torch.full((M, N), 3.0, dtype=torch.bfloat16, device=DEV)
        yv.linear(ad, wd, bd, o, flags=flags)
        if kind != "gelu":
            assert torch.equal(o.cpu().float(), got)
    finally:
        yv.set_option("linear_variant", 1)
        yv.set_option("linear_p8_rows", 0)


def test_linear_persistent_8phase_race_screen(yv):
    """Short form of tools/p8_race_screen.py (5,120 runs, profiles/r02_p8_race_screen.txt): the persistent kernel must be bit-identical
    to the round-1 128 x 128 kernel (both sum the K steps in order) on every repetition, for every tile height, at 208 and 256
    workgroups, with a library GEMM running beside it on a second stream."""
    g = torch.Generator().manual_seed(0)
    side = torch.cuda.Stream()
    noise = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
    try:
        for (m, n, k) in ((12608, 768, 768), (6304, 2304, 768), (2048 + 37, 1536, 128)):
            a = bf(torch.randn(m, k, generator=g)).to(DEV); w = bf(torch.randn(n, k, generator=g) * 0.05).to(DEV)
            bias = torch.randn(n, generator=g).to(DEV); res0 = torch.randn(m, n, generator=g).to(DEV)
            for flags, dt in ((yv.EPI_GELU, torch.bfloat16), (yv.EPI_RES_F32, torch.float32)):
                def run(variant, cus=0, rows=0):
                    yv.set_option("linear_variant", variant); yv.set_option("linear_p8_cus", cus); yv.set_option("linear_p8_rows", rows)
                    out = res0.clone() if flags & yv.EPI_RES_F32 else torch.full((m, n), 3.0, dtype=dt, device=DEV)
                    yv.linear(a, w, bias, out, flags=flags)
                    return out
                ref = run(3)
                for cus in (0, 208):
                    for rows in (0, 128, 160, 192):
                        for r in range(3):
                            if r & 1:
                                with torch.cuda.stream(side):
                                    noise @ noise
                            assert torch.equal(run(9, cus, rows), ref), (m, n, k, flags, cus, rows, r)
        torch.cuda.synchronize()
    finally:
        yv.set_option("linear_variant", 1); yv.set_option("linear_p8_cus", 0); yv.set_option("linear_p8_rows", 0)


@pytest.mark.parametrize("M,N,K,rows", [(6304, 3072, 768, 0), (6304, 768, 3072, 0), (2048 + 37, 768, 256, 160), (6304, 768, 768, 192),
                                       (9999, 512, 128, 224)])
def test_linear_free_running_trainer_epilogues(yv, M, N, K, rows):
    """The trainer's forms of yv_linear_ex on gemm_p9_kernel against the 128 x 128 kernel (same rounding steps: bit for bit):
    f32 residual read from ANOTHER tensor (x_mid = x_in + ...), YV_EPI_SAVE_PRE (pre-activation kept next to the GELU output),
    YV_EPI_GELU_BWD (data gradient of fc2 times gelu'(saved pre-activation))."""
    g = torch.Generator().manual_seed(M + N + K)
    a = bf(torch.randn(M, K, generator=g)).to(DEV)
    w = bf(torch.randn(N, K, generator=g) * 0.05).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    xin = torch.randn(M, N, generator=g).to(DEV)
    u = bf(torch.randn(M, N, generator=g)).to(DEV)

    def run(p8, form):
        yv.set_option("linear_p8", p8); yv.set_option("linear_p8_rows", rows if p8 else 0)
        if form == "resf":
            out = torch.full((M, N), 7.0, device=DEV)
            yv.linear_ex(a, w, bias, out, flags=yv.EPI_RES_F32, res_f32=xin)
            return (out,)
        if form == "save_pre":
            out = torch.full((M, N), 3.0, dtype=torch.bfloat16, device=DEV); pre = torch.full((M, N), 5.0, dtype=torch.bfloat16, device=DEV)
            yv.linear_ex(a, w, bias, out, flags=yv.EPI_GELU | yv.EPI_SAVE_PRE, aux=pre)
            return out, pre
        out = torch.full((M, N), 3.0, dtype=torch.bfloat16, device=DEV)
        yv.linear_ex(a, w, None, out, flags=yv.EPI_GELU_BWD, aux=u)
        return (out,)
    try:
        for form in ("resf", "save_pre", "gelu_bwd"):
            ref, got = run(0, form), run(3, form)
            torch.cuda.synchronize()
            for r, o in zip(ref, got):
                assert torch.equal(r, o), (form, float((r.float() - o.float()).abs().max()))
        # the reference itself against torch on the first form
        out = run(3, "resf")[0]
        assert rel_l2(out.cpu(), xin.cpu() + a.float().cpu() @ w.float().cpu().t() + bias.cpu()) < 1e-5
    finally:
        yv.set_option("linear_p8", 3); yv.set_option("linear_p8_rows", 0)


def test_transpose_bf16_batched(yv):
    g = torch.Generator().manual_seed(5)
    src = bf(torch.randn(3 * 1000 + 64 * 3, generator=g)).to(DEV)          # three 24 x 40 matrices, 1064 elements apart (16-byte aligned)
    dst = torch.zeros_like(src)
    yv.transpose_bf16_batched(src, dst, 24, 40, 3, 1064, 1064)
    torch.cuda.synchronize()
    for b_ in range(3):
        m = src[b_ * 1064: b_ * 1064 + 960].view(24, 40)
        assert torch.equal(dst[b_ * 1064: b_ * 1064 + 960].view(40, 24), m.t())
    big = bf(torch.randn(2304, 768, generator=g)).to(DEV)
    out = torch.empty(768, 2304, dtype=torch.bfloat16, device=DEV)
    yv.transpose_bf16_batched(big, out, 2304, 768)
    assert torch.equal(out, big.t())


def test_gelu_fast_form_accuracy(yv):
    """The sigmoid-form GELU of the persistent kernel against torch's erf GELU on a dense grid of arguments (identity weight:
    the GEMM reproduces its bf16 input exactly): abs error <= 2.5e-5 (fit) + half a bf16 step of the output."""
    N = K = 256
    M = 2048
    xs = torch.linspace(-12, 12, M * N).view(M, N)
    a = bf(xs)
    w = torch.eye(N)
    o = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    yv.set_option("linear_variant", 9)
    try:
        yv.linear(a.to(DEV), bf(w).to(DEV), None, o, flags=yv.EPI_GELU)
    finally:
        yv.set_option("linear_variant", 1)
    x = a.float()
    ref = F.gelu(x.double()).float()
    err = (o.cpu().float() - ref).abs()
    bound = 3e-5 + ref.abs() * 2.0 ** -8
    assert bool((err <= bound).all()), float((err - bound).max())


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,Cin,Cout,H,k,s", [(2, 16, 32, 40, 3, 2), (1, 32, 32, 24, 3, 1), (2, 64, 128, 20, 3, 2),
                                               (3, 48, 32, 16, 1, 1), (1, 256, 256, 20, 1, 1), (2, 64, 64, 20, 3, 1),
                                               (1, 128, 16, 12, 3, 1), (2, 24, 8, 10, 3, 1)])
def test_conv_vs_torch(yv, B, Cin, Cout, H, k, s):
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H + k)
    x = bf(torch.randn(B, Cin, H, H, generator=g))
    w = bf(torch.randn(Cout, Cin, k, k, generator=g) * math.sqrt(2.0 / (Cin * k * k)))
    b = torch.randn(Cout, generator=g) * 0.1
    ref = F.silu(F.conv2d(x.float(), w.float(), b, stride=s, padding=k // 2))
    Ho = ref.shape[-1]
    xin = _nhwc(x).to(DEV)
    wk = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous().to(DEV)
    out = torch.zeros(B, Ho, Ho, Cout + 8, dtype=torch.bfloat16, device=DEV)
    yv.conv2d(yv.view(xin, 0, Cin), None, B, Ho, Ho, k, s, wk, b.to(DEV), out, 8, yv.EPI_SILU)
    got = out[..., 8:].permute(0, 3, 1, 2).float().cpu()
    assert rel_l2(got, ref) < 4e-3
    assert float(out[..., :8].abs().sum()) == 0                     # channel offset respected
    # f32 output without activation (detect head tail)
    out32 = torch.zeros(B, Ho, Ho, Cout, dtype=torch.float32, device=DEV)
    yv.conv2d(yv.view(xin, 0, Cin), None, B, Ho, Ho, k, s, wk, b.to(DEV), out32, 0, yv.EPI_OUT_F32)
    ref2 = F.conv2d(x.float(), w.float(), b, stride=s, padding=k // 2)
    assert torch.allclose(out32.permute(0, 3, 1, 2).cpu(), ref2, atol=2e-3, rtol=1e-4)


def test_conv_concat_upsample_residual(yv):
    g = torch.Generator().manual_seed(9)
    B, H = 2, 20
    small = bf(torch.randn(B, 32, H // 2, H // 2, generator=g)); skip = bf(torch.randn(B, 16, H, H, generator=g))
    w = bf(torch.randn(24, 48, 1, 1, generator=g) * 0.2); b = torch.randn(24, generator=g) * 0.1
    cat = torch.cat([F.interpolate(small.float(), scale_factor=2, mode="nearest"), skip.float()], 1)
    ref = F.silu(F.conv2d(cat, w.float(), b))
    out = torch.zeros(B, H, H, 24, dtype=torch.bfloat16, device=DEV)
    sm = _nhwc(small).to(DEV); sk = _nhwc(skip).to(DEV)
    yv.conv2d(yv.view(sm, 0, 32, up=1), yv.view(sk, 0, 16), B, H, H, 1, 1, w.reshape(24, 48).contiguous().to(DEV),
              b.to(DEV), out, 0, yv.EPI_SILU)
    assert rel_l2(out.permute(0, 3, 1, 2).float().cpu(), ref) < 4e-3
    # bottleneck shortcut: y = x + SiLU(conv(x)) reading / writing channel slices of one buffer
    buf = bf(torch.randn(B, H, H, 64, generator=g)).to(DEV)
    w3 = bf(torch.randn(16, 16, 3, 3, generator=g) * 0.1); b3 = torch.randn(16, generator=g) * 0.1
    xin = buf[..., 16:32].permute(0, 3, 1, 2).float().cpu()
    ref3 = xin + F.silu(F.conv2d(xin, w3.float(), b3, padding=1))
    yv.conv2d(yv.view(buf, 16, 16), None, B, H, H, 3, 1, w3.permute(0, 2, 3, 1).reshape(16, 144).contiguous().to(DEV),
              b3.to(DEV), buf, 32, yv.EPI_SILU | yv.EPI_RES_BF16, res=buf, res_c_off=16)
    assert rel_l2(buf[..., 32:48].permute(0, 3, 1, 2).float().cpu(), ref3) < 4e-3


def test_conv_sources_beyond_2gb_are_subbatched(yv):
    """The conv kernel addresses a source with 32-bit byte offsets; a batch whose source tensor passes 2 GB (here 40 images of
    320 x 320 pixels at a pixel stride of 336 channels = 2.75 GB) is taken in sub-batches by the host: results must equal the same
    convolution over a compact copy of the 16 channels that are read (3 x 3 / stride 2 and 1 x 1 with a bf16 shortcut)."""
    B, H, ld, c = 40, 320, 336, 16
    g = torch.Generator(device=DEV).manual_seed(3)
    big = torch.randn(B, H, H, ld, generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    assert big.numel() * 2 > 2 ** 31
    small = big[..., 8:8 + c].contiguous()
    gc = torch.Generator().manual_seed(4)
    w3 = bf(torch.randn(32, 3, 3, c, generator=gc) * 0.1).reshape(32, 9 * c).contiguous().to(DEV); b3 = (torch.randn(32, generator=gc) * 0.1).to(DEV)
    w1 = bf(torch.randn(16, c, generator=gc) * 0.2).contiguous().to(DEV); b1 = (torch.randn(16, generator=gc) * 0.1).to(DEV)
    for (k, s_, w, b_, co) in ((3, 2, w3, b3, 32), (1, 1, w1, b1, 16)):
        Ho = H // s_
        out_big = torch.zeros(B, Ho, Ho, co, dtype=torch.bfloat16, device=DEV)
        out_small = torch.zeros_like(out_big)
        res = (big, 8) if k == 1 else (None, 0)
        res_s = (small, 0) if k == 1 else (None, 0)
        fl = yv.EPI_SILU | (yv.EPI_RES_BF16 if k == 1 else 0)
        yv.conv2d(yv.view(big, 8, c), None, B, Ho, Ho, k, s_, w, b_, out_big, 0, fl, res=res[0], res_c_off=res[1])
        yv.conv2d(yv.view(small, 0, c), None, B, Ho, Ho, k, s_, w, b_, out_small, 0, fl, res=res_s[0], res_c_off=res_s[1])
        assert torch.equal(out_big, out_small), k
        assert float(out_big[-1].float().abs().sum()) > 0


def test_layernorm(yv):
    g = torch.Generator().manual_seed(1)
    for D, rows in ((768, 197 * 2), (1024, 33), (128, 10)):
        x = torch.randn(rows, D, generator=g) * 3 + 0.5
        ga = 1 + 0.1 * torch.randn(D, generator=g); be = 0.1 * torch.randn(D, generator=g)
        ref = F.layer_norm(x, (D,), ga, be, eps=1e-6)
        y = torch.zeros(rows, D, dtype=torch.bfloat16, device=DEV)
        yv.layernorm(x.to(DEV), ga.to(DEV), be.to(DEV), y, rows, D, D, D)
        # f32 statistics; error = bf16 rounding of the output (|y| <~ 4 -> 2^-7 abs)
        assert torch.allclose(y.cpu().float(), ref, atol=2e-2, rtol=8e-3)
        assert rel_l2(y.cpu().float(), ref) < 3e-3
    # strided rows (cls token only) + dynamic count
    x = torch.randn(6 * 5, 128, generator=g)
    y = torch.zeros(6, 128, dtype=torch.bfloat16, device=DEV)
    cnt = torch.tensor([4], dtype=torch.int32, device=DEV)
    yv.layernorm(x.to(DEV), torch.ones(128, device=DEV), torch.zeros(128, device=DEV), y, 6, 128, 5 * 128, 128,
                 count_dev=cnt, rows_per_count=1)
    ref = F.layer_norm(x.view(6, 5, 128)[:, 0], (128,), eps=1e-6)
    assert rel_l2(y.cpu().float()[:4], ref[:4]) < 3e-3 and float(y[4:].float().abs().sum()) == 0


@pytest.mark.parametrize("R,N,H", [(3, 197, 12), (2, 5, 2), (2, 50, 2), (1, 256, 3), (2, 33, 1), (1, 64, 2),
                                   (2, 785, 3), (1, 257, 2), (1, 512, 1), (1, 1000, 2)])
def test_attention(yv, R, N, H):
    g = torch.Generator().manual_seed(R * 7 + N)
    D = H * 64
    qkv = bf(torch.randn(R * N, 3 * D, generator=g) * 1.5)
    t = qkv.float().view(R, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    att = ((t[0] * 0.125) @ t[1].transpose(-2, -1)).softmax(-1)
    ref = (att @ t[2]).transpose(1, 2).reshape(R * N, D)
    out = torch.zeros(R * N, D, dtype=torch.bfloat16, device=DEV)
    yv.attention(qkv.to(DEV), R, N, H, out)
    # P is rounded to bf16 before P.V and the output to bf16: 2^-8 relative each
    assert rel_l2(out.cpu().float(), ref) < 8e-3
    assert torch.allclose(out.cpu().float(), ref, atol=3e-2, rtol=2e-2)


def test_attention_softmax_spike(yv):
    # one key dominates one query row (forces large max subtraction); result must stay exact-ish
    R, N, H = 1, 197, 1
    g = torch.Generator().manual_seed(5)
    qkv = bf(torch.randn(N, 192, generator=g))
    qkv[7, :64] = 8.0; qkv[100, 64:128] = 8.0
    t = qkv.float().view(1, N, 3, 1, 64).permute(2, 0, 3, 1, 4)
    ref = (((t[0] * 0.125) @ t[1].transpose(-2, -1)).softmax(-1) @ t[2]).transpose(1, 2).reshape(N, 64)
    out = torch.zeros(N, 64, dtype=torch.bfloat16, device=DEV)
    yv.attention(qkv.to(DEV), R, N, H, out)
    assert torch.allclose(out.cpu().float(), ref, atol=3e-2, rtol=2e-2)
    assert torch.allclose(out.cpu().float()[7], qkv.float()[100, 128:], atol=2e-2)


def test_wrapper_head_golden(yv, golden_dir):
    import numpy as np
    z = np.load(golden_dir + "/G6_wrapper.npz")
    feats = torch.from_numpy(z["feats"]); exp = torch.from_numpy(z["out"])
    R = feats.shape[0]
    fpad = torch.zeros(R, 1024); fpad[:, :1000] = feats
    logits = torch.zeros(R, 5, device=DEV); labels = torch.zeros(R, dtype=torch.int32, device=DEV)
    w = {k: torch.from_numpy(z[k]).to(DEV) for k in ("fc__1__weight", "fc__1__bias", "fc__3__weight", "fc__3__bias")}
    w["fc__1__weight"] = w["fc__1__weight"].t().contiguous()          # the kernel takes fc.1.weight transposed
    yv.wrapper_head(fpad.to(DEV), w["fc__1__weight"], w["fc__1__bias"], w["fc__3__weight"], w["fc__3__bias"], R, 5,
                    logits, labels)
    assert torch.allclose(logits.cpu(), exp, atol=1e-5, rtol=1e-5)          # f32, summation order only
    assert labels.cpu().tolist() == exp.argmax(1).tolist()
    # ensemble: mean of two identical models == the model
    yv.wrapper_head(fpad.to(DEV), w["fc__1__weight"], w["fc__1__bias"], w["fc__3__weight"], w["fc__3__bias"], R, 5,
                    logits, labels, scale=0.5, accumulate=False)
    yv.wrapper_head(fpad.to(DEV), w["fc__1__weight"], w["fc__1__bias"], w["fc__3__weight"], w["fc__3__bias"], R, 5,
                    logits, labels, scale=0.5, accumulate=True)
    assert torch.allclose(logits.cpu(), exp, atol=1e-5, rtol=1e-5)


def test_sppf_pool(yv):
    g = torch.Generator().manual_seed(2)
    B, H, c = 2, 20, 16
    buf = torch.zeros(B, H, H, 4 * c, dtype=torch.bfloat16)
    x = bf(torch.randn(B, H, H, c, generator=g)); buf[..., :c] = x
    d = buf.to(DEV)
    yv.sppf_pool(d, c)
    y = x.float().permute(0, 3, 1, 2)
    for i in range(1, 4):
        y = F.max_pool2d(y, 5, 1, 2)
        assert torch.equal(d[..., i * c:(i + 1) * c].float().cpu(), y.permute(0, 2, 3, 1))    # max is exact
    assert torch.equal(d[..., :c].cpu(), x)


@pytest.mark.parametrize("B,H,W,cout,ld", [(2, 64, 64, 16, 16), (2, 48, 256, 16, 16), (1, 130, 128, 32, 40), (3, 32, 384, 48, 48),
                                           (1, 640, 640, 16, 16)])
def test_stem_conv(yv, B, H, W, cout, ld):
    """Both forms of the stem: scalar (any even size) and the matrix-pipe form taken when W % 128 == 0 (integer pixels against
    w / 255 in two bf16 halves): same tolerance, borders (top row / left column padding) included; channels past cout untouched."""
    g = torch.Generator().manual_seed(8 + W + cout)
    img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    img[:, 0] = 255; img[:, :, 0] = 254                              # loud borders: a wrong padding rule shows
    w = torch.randn(cout, 3, 3, 3, generator=g) * 0.3; b = torch.randn(cout, generator=g) * 0.1
    ref = F.silu(F.conv2d(img.permute(0, 3, 1, 2).float() / 255.0, w, b, stride=2, padding=1))
    out = torch.full((B, H // 2, W // 2, ld), 7.0, dtype=torch.bfloat16, device=DEV)
    yv.stem_conv(img.to(DEV), w.permute(2, 3, 1, 0).reshape(27, cout).contiguous().to(DEV), b.to(DEV), out)
    got = out[..., :cout].permute(0, 3, 1, 2).float().cpu()
    assert rel_l2(got, ref) < 3e-3
    err = (got - ref).abs()
    assert bool((err <= 2e-3 + ref.abs() * 2.0 ** -7).all()), float(err.max())
    assert bool((out[..., cout:] == 7.0).all())
