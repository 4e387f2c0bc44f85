"""GPU parity of the backward-pass kernels against torch autograd on the CPU (fp32).  Inputs are
bf16-representable; tolerances reflect bf16 operands/outputs."""
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


def test_transpose_and_cast(yv):
    g = torch.Generator().manual_seed(0)
    x = bf(torch.randn(197 * 3, 200, generator=g))
    ld = (x.shape[0] + 63) // 64 * 64
    out = torch.full((200, ld), 7.0, dtype=torch.bfloat16, device=DEV)
    yv.transpose_bf16(x.to(DEV), out)
    assert torch.equal(out[:, :x.shape[0]].cpu(), x.t()) and float(out[:, x.shape[0]:].float().abs().sum()) == 0
    w = torch.randn(130, 72, generator=g)
    wb = torch.zeros(130, 72, dtype=torch.bfloat16, device=DEV); wt = torch.zeros(72, 192, dtype=torch.bfloat16, device=DEV)
    yv.cast_weights(w.to(DEV), wb, wt)
    assert torch.equal(wb.cpu(), bf(w)) and torch.equal(wt[:, :130].cpu(), bf(w).t())
    x32 = torch.randn(600, 300, generator=g)
    y = torch.zeros(600, 300, dtype=torch.bfloat16, device=DEV); cs = torch.zeros(300, device=DEV)
    ws = torch.zeros(yv.lib.yv_colsum_ws_floats(600, 300), device=DEV)
    yv.cast_colsum(x32.to(DEV), y, cs, ws)
    assert torch.equal(y.cpu(), bf(x32)) and torch.allclose(cs.cpu(), x32.sum(0), atol=1e-3, rtol=1e-5)
    cs2 = torch.zeros(300, device=DEV)
    yv.colsum_bf16(y, cs2, ws)
    assert torch.allclose(cs2.cpu(), bf(x32).float().sum(0), atol=1e-3, rtol=1e-5)


def test_linear_training_epilogues(yv):
    g = torch.Generator().manual_seed(1)
    M, N, K = 300, 256, 128
    a = bf(torch.randn(M, K, generator=g)); w = bf(torch.randn(N, K, generator=g) * 0.1); b = torch.randn(N, generator=g)
    lin = a.float() @ w.float().t() + b
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV); pre = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    yv.linear_ex(a.to(DEV), w.to(DEV), b.to(DEV), out, flags=yv.EPI_GELU | yv.EPI_SAVE_PRE, aux=pre)
    assert rel_l2(pre.cpu().float(), lin) < 3e-3 and rel_l2(out.cpu().float(), F.gelu(lin)) < 4e-3
    # GELU backward: out = (dY . W^T-copy) * gelu'(u)
    u = bf(torch.randn(M, N, generator=g) * 2)
    ut = u.float().clone().requires_grad_(True)
    F.gelu(ut).backward(lin)                                     # grad = lin * gelu'(u)
    dg = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    yv.linear_ex(a.to(DEV), w.to(DEV), b.to(DEV), dg, flags=yv.EPI_GELU_BWD, aux=u.to(DEV))
    assert rel_l2(dg.cpu().float(), ut.grad) < 6e-3
    # separate residual source
    res = torch.randn(M, N, generator=g)
    o32 = torch.zeros(M, N, device=DEV)
    yv.linear_ex(a.to(DEV), w.to(DEV), b.to(DEV), o32, flags=yv.EPI_RES_F32, res_f32=res.to(DEV))
    assert torch.allclose(o32.cpu(), res + lin, atol=1e-3, rtol=1e-4)


def test_layernorm_bwd(yv):
    g = torch.Generator().manual_seed(2)
    for rows, D in ((197 * 2, 768), (70, 128)):
        x = (torch.randn(rows, D, generator=g) * 2 + 0.3).requires_grad_(True)
        ga = (1 + 0.1 * torch.randn(D, generator=g)).requires_grad_(True); be = torch.zeros(D, requires_grad=True)
        dy = bf(torch.randn(rows, D, generator=g))
        F.layer_norm(x, (D,), ga, be, eps=1e-6).backward(dy.float())
        dx0 = torch.randn(rows, D, generator=g)
        dx = dx0.clone().to(DEV); dga = torch.zeros(D, device=DEV); dbe = torch.zeros(D, device=DEV)
        ws = torch.zeros(yv.lib.yv_layernorm_bwd_ws_floats(rows, D), device=DEV)
        yv.layernorm_bwd(x.detach().to(DEV), D, ga.detach().to(DEV), dy.to(DEV), D, rows, D, dx, D, dga, dbe, ws)
        assert torch.allclose(dx.cpu() - dx0, x.grad, atol=2e-4, rtol=1e-3)
        assert torch.allclose(dga.cpu(), ga.grad, atol=2e-3, rtol=1e-3) and torch.allclose(dbe.cpu(), be.grad, atol=2e-3, rtol=1e-3)


@pytest.mark.parametrize("R,N,H", [(2, 197, 3), (1, 50, 2), (2, 33, 1), (1, 256, 2), (1, 785, 2), (1, 257, 1), (1, 600, 1)])
def test_attention_bwd(yv, R, N, H):
    g = torch.Generator().manual_seed(R + N)
    D = H * 64
    qkv = bf(torch.randn(R * N, 3 * D, generator=g))
    do = bf(torch.randn(R * N, D, generator=g))
    t = qkv.float().clone().requires_grad_(True)
    tt = t.view(R, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    att = ((tt[0] * 0.125) @ tt[1].transpose(-2, -1)).softmax(-1)
    ref_o = (att @ tt[2]).transpose(1, 2).reshape(R * N, D)
    ref_o.backward(do.float())
    out = torch.zeros(R * N, D, dtype=torch.bfloat16, device=DEV); lse = torch.zeros(R * H * N, device=DEV)
    yv.attention_train(qkv.to(DEV), R, N, H, out, lse)
    assert rel_l2(out.cpu().float(), ref_o.detach()) < 8e-3
    dqkv = torch.zeros(R * N, 3 * D, dtype=torch.bfloat16, device=DEV); dws = torch.zeros(R * H * N, device=DEV)
    yv.attention_bwd(qkv.to(DEV), out, do.to(DEV), lse, R, N, H, dqkv, dws)
    got = dqkv.cpu().float()
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert rel_l2(got[:, sl], t.grad[:, sl]) < 2e-2, name


def test_head_bwd(yv):
    g = torch.Generator().manual_seed(4)
    R, nc = 9, 5
    f = torch.randn(R, 1000, generator=g, requires_grad=True)
    w1 = (torch.randn(128, 1000, generator=g) * 0.05).requires_grad_(True); b1 = (torch.randn(128, generator=g) * 0.1).requires_grad_(True)
    w2 = (torch.randn(nc, 128, generator=g) * 0.2).requires_grad_(True); b2 = torch.zeros(nc, requires_grad=True)
    dl = torch.randn(R, nc, generator=g)
    (F.linear(F.relu(F.linear(F.relu(f), w1, b1)), w2, b2)).backward(dl)
    fp = torch.zeros(R, 1024); fp[:, :1000] = f.detach()
    z = lambda *s: torch.zeros(*s, device=DEV)
    dw1, db1, dw2, db2 = z(128, 1000), z(128), z(nc, 128), z(nc)
    dfe = torch.zeros(R, 1024, dtype=torch.bfloat16, device=DEV)
    yv.head_bwd(fp.to(DEV), w1.detach().t().contiguous().to(DEV), b1.detach().to(DEV), w2.detach().to(DEV), dl.to(DEV), R, nc,
                dw1, db1, dw2, db2, dfe, z(2 * R * 128))
    assert torch.allclose(dw1.cpu(), w1.grad, atol=1e-5, rtol=1e-4) and torch.allclose(db1.cpu(), b1.grad, atol=1e-5, rtol=1e-4)
    assert torch.allclose(dw2.cpu(), w2.grad, atol=1e-5, rtol=1e-4) and torch.allclose(db2.cpu(), b2.grad, atol=1e-5, rtol=1e-4)
    assert rel_l2(dfe.cpu().float()[:, :1000], f.grad) < 4e-3 and float(dfe[:, 1000:].float().abs().sum()) == 0


# ------------------------------------------------------------------------------------------ full trainer
def _oracle_grads(sd, x, labels, name):
    from oracle import train as ot, vit as ov
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits = ov.wrapper_forward(p, x, name)
    loss = ot.build_loss(logits, F.one_hot(labels.long(), 5).float())
    loss.backward()
    return loss.detach(), logits.detach(), {k: v.grad for k, v in p.items()}


@pytest.mark.parametrize("name,R", [("vit_tiny_test", 3), ("vit_tiny_test", 33), ("vit_tiny8_test", 2)])
def test_trainer_gradients_vs_autograd(yv, name, R):
    from oracle import boxes as ob, vit as ov
    from yvhip.training import VitTrainer
    from test_gpu_configs import _relu_free_head      # ReLU coin flips of the wrapper head taken out (see its docstring)
    sd = _relu_free_head(ov.init_wrapper_state(name, seed=21))
    g = torch.Generator().manual_seed(R)
    x = (torch.rand(R, 3, 224, 224, generator=g) * 2 - 1).to(torch.bfloat16).float()
    labels = torch.randint(0, 5, (R,), generator=g, dtype=torch.int32)
    ref_loss, ref_logits, ref = _oracle_grads(sd, x, labels, name)
    tr = VitTrainer(sd, name, 5)
    pm = torch.cat([torch.from_numpy(ob.patchify(x[r].numpy(), tr.P_)) for r in range(R)]).to(torch.bfloat16).to(DEV)
    logits = tr.forward(pm, R)
    loss = tr.backward(pm, labels.to(DEV), R)
    torch.cuda.synchronize()
    assert rel_l2(logits.cpu(), ref_logits) < 2e-2
    assert abs(float(loss[0]) - float(ref_loss)) < 2e-2 * abs(float(ref_loss))
    got = tr.grad_dict()
    worst = {}
    for k, v in ref.items():
        worst[k] = rel_l2(got[k].cpu(), v)
    ranked = sorted(worst.items(), key=lambda kv: -kv[1])
    print(f"{name} R={R}: gradient rel-L2 vs fp32 autograd, worst: " + ", ".join(f"{k} {e:.3f}" for k, e in ranked[:4]))
    bad = {k: e for k, e in worst.items() if e > 2e-2}
    # what remains without the ReLU coin flips: bf16 storage noise of activations and activation gradients (measured <= 0.7 %)
    assert not bad, bad


def test_trainer_two_steps_follow_sgd(yv):
    from oracle import boxes as ob, train as ot, vit as ov
    from yvhip.training import VitTrainer
    name, R = "vit_tiny_test", 16
    sd = ov.init_wrapper_state(name, seed=5)
    g = torch.Generator().manual_seed(9)
    x = (torch.rand(R, 3, 224, 224, generator=g) * 2 - 1).to(torch.bfloat16).float()
    labels = torch.randint(0, 5, (R,), generator=g, dtype=torch.int32)
    tr = VitTrainer(sd, name, 5)
    pm = torch.cat([torch.from_numpy(ob.patchify(x[r].numpy(), tr.P_)) for r in range(R)]).to(torch.bfloat16).to(DEV)
    # oracle: the reference optimizer (utils/trainClass.py:442-443) on the fp32 model, LR from the cosine schedule
    params = {k: v.clone() for k, v in sd.items()}
    bufs = {k: None for k in sd}
    losses, ref_losses = [], []
    for step in range(2):
        lr = 0.004 * ot.cosine_lr(step, 10, 1.0)
        loss, _ = tr.step(pm, labels.to(DEV), lr)
        losses.append(float(loss[0]))
        rl, _, gr = _oracle_grads(params, x, labels, name)
        ref_losses.append(float(rl))
        for k in params:
            params[k], bufs[k] = ot.sgd_step(params[k], gr[k], bufs[k], lr)
    torch.cuda.synchronize()
    assert abs(losses[0] - ref_losses[0]) < 2e-2 * abs(ref_losses[0]) and abs(losses[1] - ref_losses[1]) < 3e-2 * abs(ref_losses[1])
    assert losses[1] < losses[0]                                   # the step descends
    new = tr.state_dict()
    # parameter UPDATE (delta) matches the oracle's update
    for k in ("fc.3.weight", "fc.1.weight", "model.head.weight", "model.blocks.1.mlp.fc2.weight", "model.blocks.0.attn.qkv.weight",
              "model.patch_embed.proj.weight", "model.pos_embed", "model.norm.weight"):
        d_ref = params[k] - sd[k]
        d_got = new[k].cpu() - sd[k]
        assert rel_l2(d_got, d_ref) < 1.2e-1, k            # two accumulated bf16-noise gradients (see the table tool)


@pytest.mark.parametrize("T,N,K", [(64, 128, 128), (6336, 2304, 768), (448, 1000, 768), (640, 768, 3072), (128, 72, 200)])
def test_wgrad_transposing_gemm(yv, T, N, K):
    """dW = dY^T . X with hardware-transposed LDS reads: small-integer operands make the f32 result EXACT, so a
    wrong lane map / swizzle / token order cannot hide behind a tolerance; asymmetric operands catch transposes."""
    g = torch.Generator().manual_seed(T + N)
    live = T - 37 if T > 64 else T                                   # zero-padded tail rows
    dy = torch.zeros(T, N); dy[:live] = torch.randint(-3, 4, (live, N), generator=g).float()
    x = torch.zeros(T, K); x[:live] = torch.randint(-2, 3, (live, K), generator=g).float()
    ref = dy.t() @ x
    dw = torch.full((N, K), 7.0, device=DEV)
    yv.wgrad(dy.to(torch.bfloat16).to(DEV), x.to(torch.bfloat16).to(DEV), dw)
    assert torch.equal(dw.cpu(), ref)
    # column-sliced dY (the backbone-head gradient uses the first 1000 of 1024 columns)
    if N == 1000:
        dyp = torch.zeros(T, 1024, dtype=torch.bfloat16); dyp[:, :1000] = dy.to(torch.bfloat16)
        dw2 = torch.zeros(N, K, device=DEV)
        yv.wgrad(dyp.to(DEV)[:, :1000], x.to(torch.bfloat16).to(DEV), dw2)
        assert torch.equal(dw2.cpu(), ref)


@pytest.mark.parametrize("M,N,K", [(300, 768, 2304), (6304, 3072, 768), (64, 128, 64), (33, 768, 1024)])
def test_linear_nn_reduction_major_weight(yv, M, N, K):
    """out = A . W with W stored (K,N): exact on small integers (catches operand-map / swizzle errors)."""
    g = torch.Generator().manual_seed(M + N)
    a = torch.randint(-3, 4, (M, K), generator=g).float(); w = torch.randint(-2, 3, (K, N), generator=g).float()
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    yv.linear_nn(a.to(torch.bfloat16).to(DEV), w.to(torch.bfloat16).to(DEV), out)
    ref = a @ w
    assert torch.equal(out.cpu().float(), ref.to(torch.bfloat16).float())
