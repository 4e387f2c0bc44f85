"""ViT fine-tune step on the HIP path (reference loop utils/trainClass.py:374-420).

One `VitTrainer.step(patches, labels, lr)` = forward (activations kept) -> build_loss (fused kernel) ->
backward -> [data-parallel: bucketed gradient SUM all-reduce over RCCL, overlapped with the rest of
backward] -> SGD(momentum 0.9, weight_decay 1e-3) on fp32 master weights -> bf16 working copies refreshed.

Memory layout (sized for 288 GB of HBM: nothing is recomputed except softmax probabilities):
  * parameters, gradients and momentum are three FLAT fp32 buffers in state-dict order (64-float aligned
    slots): the optimizer is ONE kernel launch and all-reduce buckets are plain slices;
  * the bf16 working weights are ONE flat mirror of the master buffer, written by the SGD kernel itself;
  * dgrad (dY . W) and wgrad (dY^T . X) read their reduction-major operands with the gfx950 transposing LDS
    read, so neither W^T copies nor transposed activations exist; wgrad operands carry 64-row zero padding.
Gradients arrive in exactly the reverse of the flat order (head first, patch-embed last), so a bucket is
complete - and its all-reduce can start - as soon as backward has passed its lowest offset.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import (EPI_GELU, EPI_GELU_BWD, EPI_OUT_F32, EPI_POSEMB, EPI_RES_F32, EPI_SAVE_PRE, YvError, attention_bwd,
               attention_train, cast_colsum, cls_rows, colsum_bf16, head_bwd, layernorm, layernorm_bwd, lib,
               linear, linear_ex, linear_nn, loss_fwd_bwd, require_gpu, sgd_step, token_reduce, transpose_bf16_batched, wgrad,
               wrapper_head)
from .engines import vit_cfg


def _r64(n: int) -> int:
    return (n + 63) // 64 * 64


class VitTrainer:
    def __init__(self, state: Dict[str, torch.Tensor], name: str, num_classes: int = 5, img: int = 224,
                 device: str = "cuda:0", momentum: float = 0.9, weight_decay: float = 1e-3,
                 bucket_mb: float = 32.0):
        require_gpu()
        self.P_, self.D, self.L, self.H = vit_cfg(name)
        self.name, self.nc, self.img, self.dev = name, num_classes, img, torch.device(device)
        self.tok = (img // self.P_) ** 2
        self.N = self.tok + 1
        self.momentum, self.wd = momentum, weight_decay
        self.steps = 0
        # ---- flat fp32 parameter / gradient / momentum buffers --------------------------------
        self.names: List[str] = list(state.keys())
        self.shapes = {k: tuple(state[k].shape) for k in self.names}
        self.off: Dict[str, int] = {}
        o = 0
        for k in self.names:
            self.off[k] = o
            o += _r64(state[k].numel())
        self.total = o
        z = lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=self.dev)
        self.P, self.G, self.Mo = z(o), z(o), z(o)
        for k in self.names:
            self.p(k).copy_(state[k].to(self.dev, torch.float32))
        # ---- bf16 working copies of the GEMM weights: ONE flat bf16 mirror of P (same offsets), refreshed by the
        # SGD kernel itself; a GEMM weight is a view into it, used as (N,K) by forward / wgrad and, through the
        # transposing-read kernel, as the reduction-major operand of dgrad (no W^T copies)
        D = self.D
        self.P16 = torch.zeros(o, dtype=torch.bfloat16, device=self.dev)
        self.gemm_w: Dict[str, tuple] = {}                 # key -> (N, K, wb view (N,K))
        def reg(key, N, K):
            self.gemm_w[key] = (N, K, self._view16(key).reshape(N, K))
        reg("model.patch_embed.proj.weight", D, 3 * self.P_ * self.P_)
        for i in range(self.L):
            b = f"model.blocks.{i}."
            reg(b + "attn.qkv.weight", 3 * D, D); reg(b + "attn.proj.weight", D, D)
            reg(b + "mlp.fc1.weight", 4 * D, D); reg(b + "mlp.fc2.weight", D, 4 * D)
        reg("model.head.weight", 1000, D)
        self.w_head_pad = torch.zeros((1024, D), dtype=torch.bfloat16, device=self.dev)   # dgrad reduces over 1024
        self.b_head_pad = z(1024)
        # ---- TRANSPOSED bf16 mirror of the block linears (same offsets; (K, N) row-major): the data gradient dX = dY . W is then
        # an ordinary linear on W^T and runs on the persistent forward GEMM (round 3; the transposing-read kernel on the master
        # layout ran at 0.12 of the MFMA roof and was a quarter of the step).  Refreshed after every optimizer step by four
        # batched transposes (one per linear of a block, batch = depth): 170 MB read + 170 MB written, ~1 % of a step.
        self.P16T = torch.zeros(o, dtype=torch.bfloat16, device=self.dev)
        self.blk_stride = (self.off["model.blocks.1.attn.qkv.weight"] - self.off["model.blocks.0.attn.qkv.weight"]) if self.L > 1 else 0
        for i in range(1, self.L):
            for w in ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight"):
                if self.off[f"model.blocks.{i}.{w}"] - self.off[f"model.blocks.0.{w}"] != i * self.blk_stride:
                    raise YvError("state dict: blocks are not laid out with one stride")
        self.P16.copy_(self.P)                               # initial cast (plumbing); afterwards the SGD kernel mirrors
        self.refresh_working_copies()
        self._bufs: Dict[int, dict] = {}
        self.s_w = None                                      # side stream of the weight gradients (backward)
        from .dist import BucketReducer
        self.reducer = BucketReducer(self.G, int(bucket_mb * 1024 * 1024 / 4))

    # ---- views ----------------------------------------------------------------------------------
    def _view(self, flat, k):
        o = self.off[k]
        n = 1
        for d in self.shapes[k]:
            n *= d
        return flat[o:o + n].view(self.shapes[k])

    def _view16(self, k):
        o = self.off[k]
        n = 1
        for d in self.shapes[k]:
            n *= d
        return self.P16[o:o + n]

    def p(self, k):
        return self._view(self.P, k)

    def g(self, k):
        return self._view(self.G, k)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: self.p(k).detach().clone() for k in self.names}

    def grad_dict(self) -> Dict[str, torch.Tensor]:
        return {k: self.g(k).detach().clone() for k in self.names}

    def refresh_working_copies(self):
        """Only the 1000 -> 1024 padded head copies need touching: everything else is a view of the mirror."""
        self.w_head_pad[:1000].copy_(self.gemm_w["model.head.weight"][2])
        self.b_head_pad[:1000].copy_(self.p("model.head.bias"))
        for w in ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight"):
            key = "model.blocks.0." + w
            N, K = self.gemm_w[key][0], self.gemm_w[key][1]
            o = self.off[key]
            transpose_bf16_batched(self.P16[o:], self.P16T[o:], N, K, self.L, self.blk_stride, self.blk_stride)

    def wt(self, key: str) -> torch.Tensor:
        """W^T of a block linear, (K, N) row-major bf16 (view of the transposed mirror)."""
        N, K = self.gemm_w[key][0], self.gemm_w[key][1]
        o = self.off[key]
        return self.P16T[o:o + N * K].view(K, N)

    # ---- buffers ----------------------------------------------------------------------------------
    def _buffers(self, R: int) -> dict:
        if R in self._bufs:
            return self._bufs[R]
        dev, D, N, L = self.dev, self.D, self.N, self.L
        M, Mp = R * N, _r64(R * N)
        f32 = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        b16 = lambda *s: torch.zeros(s, dtype=torch.bfloat16, device=dev)
        Rp, Tp = _r64(R), _r64(R * self.tok)
        # operands of a weight gradient are allocated with their token dimension padded to 64 ZERO rows (never
        # written: every producer stops at row M); "*_full" is the padded tensor, the plain name its [:M] view
        pad = lambda rows, rp, cols: b16(rp, cols)
        full = dict(h1=[pad(M, Mp, D) for _ in range(L)], o=[pad(M, Mp, D) for _ in range(L)],
                    h2=[pad(M, Mp, D) for _ in range(L)], g=[pad(M, Mp, 4 * D) for _ in range(L)],
                    dxb=pad(M, Mp, D), dwide=pad(M, Mp, 4 * D), dqkv=pad(M, Mp, 3 * D),
                    dfeats=pad(R, Rp, 1024), c=pad(R, Rp, D), dtok=pad(R * self.tok, Tp, D),
                    patches=pad(R * self.tok, Tp, 3 * self.P_ * self.P_))
        b = dict(M=M, Mp=Mp, Rp=Rp, Tp=Tp, full=full,
                 x=[f32(M, D) for _ in range(2 * L + 1)],                      # residual stream snapshots
                 h1=[t[:M] for t in full["h1"]], qkv=[b16(M, 3 * D) for _ in range(L)],
                 o=[t[:M] for t in full["o"]], lse=[f32(R * self.H * N) for _ in range(L)],
                 h2=[t[:M] for t in full["h2"]], u=[b16(M, 4 * D) for _ in range(L)], g=[t[:M] for t in full["g"]],
                 c=full["c"][:R], feats=f32(R, 1024), logits=f32(R, self.nc), labels=torch.zeros(R, dtype=torch.int32, device=dev),
                 dx=f32(M, D), dxb=full["dxb"][:M], dwide=full["dwide"][:M], dqkv=full["dqkv"][:M], dnar=b16(M, D),
                 delta=f32(R * self.H * N), dfeats=full["dfeats"][:R], dc=b16(R, D),
                 dtok=full["dtok"][:R * self.tok], dtok32=f32(R * self.tok, D), patches=full["patches"][:R * self.tok],
                 dpos=f32(N, D),
                 # gradient operands of a block's four weight gradients, double-buffered by block parity: the weight
                 # gradients of block i run on a second stream while the main stream already writes block i-1's set
                 dy=[dict(dxb_fc2=pad(M, Mp, D), dxb_proj=pad(M, Mp, D), dwide=pad(M, Mp, 4 * D), dqkv=pad(M, Mp, 3 * D),
                          done=None) for _ in range(2)],
                 ws=f32(max(int(lib.yv_colsum_ws_floats(M, 4 * D)), int(lib.yv_layernorm_bwd_ws_floats(M, D)), 2 * R * 128) + 64),
                 ws_w=f32(int(lib.yv_colsum_ws_floats(M, 4 * D)) + 64))          # scratch of the column sums on the side stream
        self._bufs[R] = b
        return b

    # ---- forward (activations kept) -------------------------------------------------------------------
    def forward(self, patches: torch.Tensor, R: int) -> torch.Tensor:
        b = self._buffers(R)
        D, N, tok, H, L = self.D, self.N, self.tok, self.H, self.L
        M = R * N
        W = lambda k: self.gemm_w[k][2]
        b["patches"].copy_(patches)                            # token-padded copy (operand of the patch-embed wgrad)
        patches = b["patches"]
        x0 = b["x"][0]
        cls_rows(self.p("model.cls_token").reshape(D), self.p("model.pos_embed").reshape(N, D), R, tok, D, x0)
        linear(patches, W("model.patch_embed.proj.weight"), self.p("model.patch_embed.proj.bias"), x0,
               flags=EPI_OUT_F32 | EPI_POSEMB, pos=self.p("model.pos_embed").reshape(N, D), tok=tok)
        for i in range(L):
            k = f"model.blocks.{i}."
            xin, xmid, xout = b["x"][2 * i], b["x"][2 * i + 1], b["x"][2 * i + 2]
            layernorm(xin, self.p(k + "norm1.weight"), self.p(k + "norm1.bias"), b["h1"][i], M, D, D, D)
            linear(b["h1"][i], W(k + "attn.qkv.weight"), self.p(k + "attn.qkv.bias"), b["qkv"][i])
            attention_train(b["qkv"][i], R, N, H, b["o"][i], b["lse"][i])
            linear_ex(b["o"][i], W(k + "attn.proj.weight"), self.p(k + "attn.proj.bias"), xmid, flags=EPI_RES_F32, res_f32=xin)
            layernorm(xmid, self.p(k + "norm2.weight"), self.p(k + "norm2.bias"), b["h2"][i], M, D, D, D)
            linear_ex(b["h2"][i], W(k + "mlp.fc1.weight"), self.p(k + "mlp.fc1.bias"), b["g"][i],
                      flags=EPI_GELU | EPI_SAVE_PRE, aux=b["u"][i])
            linear_ex(b["g"][i], W(k + "mlp.fc2.weight"), self.p(k + "mlp.fc2.bias"), xout, flags=EPI_RES_F32, res_f32=xmid)
        xf = b["x"][2 * L]
        layernorm(xf, self.p("model.norm.weight"), self.p("model.norm.bias"), b["c"], R, D, N * D, D)
        linear(b["c"], self.w_head_pad, self.b_head_pad, b["feats"], flags=EPI_OUT_F32)
        w1t = self.p("fc.1.weight").t().contiguous()
        b["w1t"] = w1t
        wrapper_head(b["feats"], w1t, self.p("fc.1.bias"), self.p("fc.3.weight"), self.p("fc.3.bias"), R, self.nc,
                     b["logits"], b["labels"])
        return b["logits"]

    # ---- backward ---------------------------------------------------------------------------------------
    def _wgrad(self, key: str, dy_full: torch.Tensor, x_full: torch.Tensor):
        """G[key] (N,K) = dy^T . x over the (64-padded, zero-tailed) token rows: transposing-read MFMA GEMM."""
        N, K = self.gemm_w[key][0], self.gemm_w[key][1]
        wgrad(dy_full, x_full, self.g(key).reshape(N, K))

    def _launch_ready_buckets(self, low_offset: int):
        """Gradients at offsets >= low_offset are final: start their all-reduce while backward continues."""
        self.reducer.ready(low_offset)

    def backward(self, patches: torch.Tensor, labels: torch.Tensor, R: int) -> torch.Tensor:
        b = self._buffers(R)
        D, N, tok, H, L = self.D, self.N, self.tok, self.H, self.L
        M = R * N
        Wm = lambda k: self.gemm_w[k][2]                    # master-layout bf16 weight (N_w, K_w)
        self.reducer.reset()
        loss, dlogits = loss_fwd_bwd(b["logits"], labels)
        # ---- Network_Wrapper.fc + backbone head -----------------------------------------------------------
        head_bwd(b["feats"], b["w1t"], self.p("fc.1.bias"), self.p("fc.3.weight"), dlogits, R, self.nc,
                 self.g("fc.1.weight"), self.g("fc.1.bias"), self.g("fc.3.weight"), self.g("fc.3.bias"), b["dfeats"], b["ws"])
        self._launch_ready_buckets(self.off["fc.1.weight"])
        colsum_bf16(b["dfeats"], b["ws"][:1024], b["ws"][1024:], rows=R)
        self.g("model.head.bias").copy_(b["ws"][:1000])
        self._wgrad("model.head.weight", b["full"]["dfeats"][:, :1000], b["full"]["c"])
        linear_nn(b["dfeats"], self.w_head_pad, b["dc"])
        b["dx"].zero_()
        layernorm_bwd(b["x"][2 * L], N * D, self.p("model.norm.weight"), b["dc"], D, R, D, b["dx"], N * D,
                      self.g("model.norm.weight"), self.g("model.norm.bias"), b["ws"])
        self._launch_ready_buckets(self.off["model.norm.weight"])
        # ---- transformer blocks, last to first -------------------------------------------------------------
        main = torch.cuda.current_stream()
        if self.s_w is None:
            self.s_w = torch.cuda.Stream()
        for i in reversed(range(L)):
            k = f"model.blocks.{i}."
            xin, xmid = b["x"][2 * i], b["x"][2 * i + 1]
            dx = b["dx"]
            S = b["dy"][i & 1]
            if S["done"] is not None:
                main.wait_event(S["done"])                     # block i+2's weight gradients have read this set
            dxb_fc2, dxb_proj, dwide, dqkv = S["dxb_fc2"][:M], S["dxb_proj"][:M], S["dwide"][:M], S["dqkv"][:M]
            # MLP branch
            cast_colsum(dx, dxb_fc2, self.g(k + "mlp.fc2.bias"), b["ws"])
            linear_ex(dxb_fc2, self.wt(k + "mlp.fc2.weight"), None, dwide, flags=EPI_GELU_BWD, aux=b["u"][i])
            linear(dwide, self.wt(k + "mlp.fc1.weight"), None, b["dnar"])
            layernorm_bwd(xmid, D, self.p(k + "norm2.weight"), b["dnar"], D, M, D, dx, D,
                          self.g(k + "norm2.weight"), self.g(k + "norm2.bias"), b["ws"])
            # attention branch
            cast_colsum(dx, dxb_proj, self.g(k + "attn.proj.bias"), b["ws"])
            linear(dxb_proj, self.wt(k + "attn.proj.weight"), None, b["dnar"])
            attention_bwd(b["qkv"][i], b["o"][i], b["dnar"], b["lse"][i], R, N, H, dqkv, b["delta"])
            linear(dqkv, self.wt(k + "attn.qkv.weight"), None, b["dnar"])
            layernorm_bwd(xin, D, self.p(k + "norm1.weight"), b["dnar"], D, M, D, dx, D,
                          self.g(k + "norm1.weight"), self.g(k + "norm1.bias"), b["ws"])
            # the block's four weight gradients and the two bias gradients that are pure column sums (fc1, qkv): nothing on the
            # data-gradient chain needs them, so they run on the side stream (own split-K workspace and column-sum scratch) under
            # the next block's chain; the gradient buckets that become final with them are launched from that stream, i.e. after them
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(self.s_w):
                self.s_w.wait_event(ev)
                colsum_bf16(dwide, self.g(k + "mlp.fc1.bias"), b["ws_w"])
                colsum_bf16(dqkv, self.g(k + "attn.qkv.bias"), b["ws_w"])
                self._wgrad(k + "mlp.fc2.weight", S["dxb_fc2"], b["full"]["g"][i])
                self._wgrad(k + "mlp.fc1.weight", S["dwide"], b["full"]["h2"][i])
                self._wgrad(k + "attn.proj.weight", S["dxb_proj"], b["full"]["o"][i])
                self._wgrad(k + "attn.qkv.weight", S["dqkv"], b["full"]["h1"][i])
                self._launch_ready_buckets(self.off[k + "norm1.weight"])
                S["done"] = torch.cuda.Event()
                S["done"].record(self.s_w)
        main.wait_stream(self.s_w)
        # ---- embeddings -----------------------------------------------------------------------------------
        token_reduce(b["dx"], R, N, D, b["dpos"])
        self.g("model.pos_embed").copy_(b["dpos"].view(1, N, D))
        self.g("model.cls_token").copy_(b["dpos"][0].view(1, 1, D))
        b["dtok32"].copy_(b["dx"].view(R, N, D)[:, 1:, :].reshape(R * tok, D))          # drop the cls rows (copy only)
        cast_colsum(b["dtok32"], b["dtok"], self.g("model.patch_embed.proj.bias"), b["ws"])
        self._wgrad("model.patch_embed.proj.weight", b["full"]["dtok"], b["full"]["patches"])
        return loss

    # ---- optimizer ----------------------------------------------------------------------------------------
    def optimizer_step(self, lr: float):
        import torch.distributed as dist
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.reducer.finish()
        sgd_step(self.P, self.G, self.Mo, lr, self.momentum, self.wd, first=self.steps == 0, grad_scale=1.0 / world,
                 mirror=self.P16)
        self.steps += 1
        self.refresh_working_copies()

    def step(self, patches: torch.Tensor, labels: torch.Tensor, lr: float):
        """One fine-tune step; `patches` (R*tok, 3*P*P) bf16 patch-major crops, `labels` (R) int32.
        Returns (loss (1,) f32 device tensor, logits (R,nc))."""
        R = labels.shape[0]
        logits = self.forward(patches, R)
        loss = self.backward(patches, labels, R)
        self.optimizer_step(lr)
        return loss, logits
