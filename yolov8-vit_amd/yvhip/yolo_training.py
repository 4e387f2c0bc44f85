"""Detector training step on the MI355X path (the ultralytics trainer behind
`YOLO(pt).train(...)`, utils/trainYolo.py:13-35; BASELINE.json configs[3]).

`YoloTrainer` owns the un-fused YOLOv8 (Conv = conv -> BatchNorm(batch statistics) -> SiLU) as flat fp32
parameter / gradient / momentum buffers plus a bf16 mirror of the conv weights, and replays fixed lists of C-ABI
launches for forward and backward.  torch only owns device memory.  Layout rules:
  * activations and their gradients are NHWC bf16 matrices (rows = B*H*W padded to a multiple of 64 with zero rows,
    channels) - C2f / SPPF / neck concats are channel slices, so "split" and "concat" are views in both directions;
  * every backward op ACCUMULATES into the gradient slice of its input (all gradient buffers are one allocation,
    zeroed once per step), which is what makes multi-consumer tensors (C2f splits, P3/P4/P5 features) correct;
  * a conv's data gradient is a conv of dz with the flipped / transposed weight (zero-inserted dz for stride 2), its
    weight gradient dz^T . im2col(x) runs on the transposing-read GEMM of the ViT trainer (yv_wgrad).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

import math

from . import (OPT_ADAMW, OPT_SGD_NESTEROV, VIEW_ADD, VIEW_COPY, VIEW_PAD, VIEW_UP2, VIEW_UP2_BWD, VIEW_ZERO_INSERT, YvError, axpby,
               blob_nhwc8, bn_act_bwd, bn_act_fwd, ema_update, optim_step,
               bn_stats, bn_ws_floats, cast_colsum, colsum_ws_floats, conv_view, conv_weight_dgrad, detect_loss,
               detect_loss_ws_bytes, im2col3, maxpool5_bwd, mview, require_gpu, sgd_step, sppf_pool, view_op, wgrad, wgrad_conv3)
from .engines import LAYER_STRIDE, REG_MAX, _c, yolo_conv_keys, yolo_layers

BN_EPS, BN_MOMENTUM = 1e-3, 0.03


def _r64(n: int) -> int:
    return (n + 63) // 64 * 64


def init_yolo_train_state(scale: str = "n", nc: int = 5, seed: int = 42) -> Dict[str, torch.Tensor]:
    """Seeded random UN-FUSED weights in the ultralytics key layout (`*.conv.weight`, `*.bn.{weight,bias,running_mean,
    running_var}`, Detect's `cv2.s.2.{weight,bias}`): there is no network on the box, so benchmarks and smoke tests
    train from this instead of a downloaded `.pt`."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for key, ci, co, k in yolo_conv_keys(scale, nc):
        w = torch.randn(co, ci, k, k, generator=g) * math.sqrt(2.0 / (ci * k * k))
        if key.endswith(".conv"):
            base = key[:-5]
            sd[base + ".conv.weight"] = w
            sd[base + ".bn.weight"] = torch.ones(co)
            sd[base + ".bn.bias"] = torch.zeros(co)
            sd[base + ".bn.running_mean"] = torch.zeros(co)
            sd[base + ".bn.running_var"] = torch.ones(co)
        else:
            sd[key + ".weight"] = w
            # ultralytics Detect.bias_init: box branch 1.0, class branch log(5 / nc / (640 / stride)^2)
            s_idx = int(key.split(".")[3])
            sd[key + ".bias"] = (torch.full((co,), 1.0) if ".cv2." in key else
                                 torch.full((co,), math.log(5 / nc / (640 / (8 << s_idx)) ** 2)))
    return sd


class _Act:
    """(B,H,W,C) bf16 activation + its gradient, stored as (rows padded to 64, C)."""

    def __init__(self, tr: "YoloTrainer", B: int, H: int, W: int, Cn: int):
        self.B, self.H, self.W, self.C = B, H, W, Cn
        self.T = B * H * W
        self.buf = torch.zeros((_r64(self.T), Cn), dtype=torch.bfloat16, device=tr.dev)
        tr._grad_specs.append(self)
        self.grad: Optional[torch.Tensor] = None

    def v(self, off: int = 0, c: Optional[int] = None):
        return mview(self.buf, off, self.C - off if c is None else c)

    def g(self, off: int = 0, c: Optional[int] = None):
        return mview(self.grad, off, self.C - off if c is None else c)


class _Block:
    """One Conv(+BN+SiLU) (bn=True) or plain biased conv (Detect's last layers)."""

    def __init__(self, tr: "YoloTrainer", key: str, cin: int, cout: int, k: int, s: int, bn: bool = True,
                 cin_real: Optional[int] = None, cout_real: Optional[int] = None):
        self.key, self.cin, self.cout, self.k, self.s, self.bn = key, cin, cout, k, s, bn
        self.cin_real = cin if cin_real is None else cin_real
        self.cout_real = cout if cout_real is None else cout_real
        self.taps = k * k
        self.w = tr._param(key + (".conv.weight" if bn else ".weight"), cout * self.taps * cin, "w")
        if bn:                                   # ultralytics groups: conv weights (decay) / BatchNorm weights / all biases
            self.gamma = tr._param(key + ".bn.weight", cout, "bnw")
            self.beta = tr._param(key + ".bn.bias", cout, "bias")
        else:
            self.bias = tr._param(key + ".bias", cout, "bias")
        tr.blocks.append(self)


class YoloTrainer:
    def __init__(self, state: Dict[str, torch.Tensor], scale: str = "n", nc: int = 5, size: int = 640, batch: int = 16,
                 lr: float = 1e-4, momentum: float = 0.937, weight_decay: float = 5e-4, device: str = "cuda:0",
                 optimizer: str = "sgd", ema: bool = False, ema_decay: float = 0.9999, ema_tau: float = 2000.0,
                 overlap_wgrad: bool = True, implicit_wgrad: bool = True):
        require_gpu()
        if size % 32:
            raise YvError("input size must be a multiple of 32")
        self.scale, self.nc, self.size, self.B, self.dev = scale, nc, size, batch, torch.device(device)
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.overlap_wgrad, self.s_w, self._pending = overlap_wgrad, None, []
        self.implicit_wgrad = implicit_wgrad               # 3x3 / stride 1 weight gradients without an im2col buffer
        if optimizer not in ("sgd", "sgd_nesterov", "adamw"):
            raise YvError("optimizer must be 'sgd', 'sgd_nesterov' or 'adamw'")
        self.optimizer, self.use_ema, self.ema_decay, self.ema_tau = optimizer, ema, ema_decay, ema_tau
        self.ncp = (nc + 7) // 8 * 8
        self.blocks: List[_Block] = []
        self._pspecs: List[Tuple[str, int, str]] = []
        self._grad_specs: List[_Act] = []
        self._build_graph()
        self._alloc_params(state)
        self._alloc_buffers()
        self.step_count = 0
        self.loss_out = torch.zeros(4, device=self.dev)
        self._loss_ws: Dict[int, torch.Tensor] = {}
        from .dist import BucketReducer
        self.reducer = BucketReducer(self.G, 1 << 62)            # 12-45 MB of gradients: one all-reduce after backward

    # ------------------------------------------------------------------ parameters
    def _param(self, name: str, n: int, group: str) -> int:
        self._pspecs.append((name, n, group))
        return len(self._pspecs) - 1

    def _alloc_params(self, state: Dict[str, torch.Tensor]):
        order = [i for g in ("w", "bnw", "bias") for i, s in enumerate(self._pspecs) if s[2] == g]
        self.off: Dict[int, Tuple[int, int]] = {}
        self.group: Dict[str, Tuple[int, int]] = {}
        pos = 0
        for i in order:
            n = (self._pspecs[i][1] + 7) // 8 * 8            # 32-byte aligned segments
            self.off[i] = (pos, self._pspecs[i][1])
            g = self._pspecs[i][2]
            lo = self.group[g][0] if g in self.group else pos
            pos += n
            self.group[g] = (lo, pos)
        self.n_weight = self.group["w"][1]
        self.n_param = pos
        host = torch.zeros(pos, dtype=torch.float32)
        for b in self.blocks:
            o, n = self.off[b.w]
            wkey = b.key + (".conv.weight" if b.bn else ".weight")
            w = state[wkey].float()
            if tuple(w.shape) != (b.cout_real, b.cin_real, b.k, b.k):
                raise YvError(f"{wkey} has shape {tuple(w.shape)}, expected {(b.cout_real, b.cin_real, b.k, b.k)}")
            wp = torch.zeros(b.cout, b.k, b.k, b.cin)
            wp[:b.cout_real, :, :, :b.cin_real] = w.permute(0, 2, 3, 1)                  # (Cout, ky, kx, Cin)
            host[o:o + n] = wp.reshape(-1)
            if b.bn:
                for pid, suffix in ((b.gamma, ".bn.weight"), (b.beta, ".bn.bias")):
                    o2, n2 = self.off[pid]
                    host[o2:o2 + n2] = state[b.key + suffix].float()
            else:
                o2, n2 = self.off[b.bias]
                host[o2:o2 + b.cout_real] = state[b.key + ".bias"].float()
        self.P = host.to(self.dev)
        self.G = torch.zeros_like(self.P)
        self.Mo = torch.zeros_like(self.P)
        self.P16 = self.P[:self.n_weight].to(torch.bfloat16)
        self.V: Optional[torch.Tensor] = None                     # AdamW second moments (allocated on first use)
        # running statistics: one flat buffer (mean | var per block) so the EMA is a single launch
        n_rs = sum(2 * b.cout for b in self.blocks if b.bn)
        rs_host = torch.zeros(n_rs)
        self._rs_off: Dict[str, int] = {}
        pos = 0
        for b in self.blocks:
            if b.bn:
                self._rs_off[b.key] = pos
                rs_host[pos:pos + b.cout] = state[b.key + ".bn.running_mean"].float()
                rs_host[pos + b.cout:pos + 2 * b.cout] = state[b.key + ".bn.running_var"].float()
                pos += 2 * b.cout
        self.RS = rs_host.to(self.dev)
        self.run_mean = {b.key: self.RS[self._rs_off[b.key]:self._rs_off[b.key] + b.cout] for b in self.blocks if b.bn}
        self.run_var = {b.key: self.RS[self._rs_off[b.key] + b.cout:self._rs_off[b.key] + 2 * b.cout] for b in self.blocks if b.bn}
        self.P_ema = self.P.clone() if self.use_ema else None
        self.RS_ema = self.RS.clone() if self.use_ema else None
        self.ema_updates = 0
        self.G_acc: Optional[torch.Tensor] = None
        self.accumulated = 0

    def p(self, pid: int) -> torch.Tensor:
        o, n = self.off[pid]
        return self.P[o:o + n]

    def gr(self, pid: int) -> torch.Tensor:
        o, n = self.off[pid]
        return self.G[o:o + n]

    def w16(self, b: _Block) -> torch.Tensor:
        o, n = self.off[b.w]
        return self.P16[o:o + n]

    def state_dict(self, ema: bool = False) -> Dict[str, torch.Tensor]:
        """Un-fused ultralytics key layout, fp32, on the host; ema=True returns the ModelEMA copy (what ultralytics saves)."""
        sd: Dict[str, torch.Tensor] = {}
        if ema and not self.use_ema:
            raise YvError("trainer was built without ema=True")
        P = (self.P_ema if ema else self.P).cpu()
        RS = (self.RS_ema if ema else self.RS).cpu()
        for b in self.blocks:
            o, n = self.off[b.w]
            w = P[o:o + n].view(b.cout, b.k, b.k, b.cin)[:b.cout_real, :, :, :b.cin_real].permute(0, 3, 1, 2).contiguous()
            if b.bn:
                sd[b.key + ".conv.weight"] = w
                for pid, suffix in ((b.gamma, ".bn.weight"), (b.beta, ".bn.bias")):
                    o2, n2 = self.off[pid]
                    sd[b.key + suffix] = P[o2:o2 + n2].clone()
                ro = self._rs_off[b.key]
                sd[b.key + ".bn.running_mean"] = RS[ro:ro + b.cout].clone()
                sd[b.key + ".bn.running_var"] = RS[ro + b.cout:ro + 2 * b.cout].clone()
            else:
                sd[b.key + ".weight"] = w
                o2, _ = self.off[b.bias]
                sd[b.key + ".bias"] = P[o2:o2 + b.cout_real].clone()
        return sd

    def grads(self) -> Dict[str, torch.Tensor]:
        """Gradients in the layout of state_dict() (tests)."""
        sd: Dict[str, torch.Tensor] = {}
        G = self.G.cpu()
        for b in self.blocks:
            o, n = self.off[b.w]
            w = G[o:o + n].view(b.cout, b.k, b.k, b.cin)[:b.cout_real, :, :, :b.cin_real].permute(0, 3, 1, 2).contiguous()
            if b.bn:
                sd[b.key + ".conv.weight"] = w
                for pid, suffix in ((b.gamma, ".bn.weight"), (b.beta, ".bn.bias")):
                    o2, n2 = self.off[pid]
                    sd[b.key + suffix] = G[o2:o2 + n2].clone()
            else:
                sd[b.key + ".weight"] = w
                o2, _ = self.off[b.bias]
                sd[b.key + ".bias"] = G[o2:o2 + b.cout_real].clone()
        return sd

    # ------------------------------------------------------------------ graph
    def _build_graph(self):
        sc, nc = self.scale, self.nc
        self.layers = yolo_layers(sc)
        self.mod: Dict[int, dict] = {}
        for idx, kind, p in self.layers:
            pre = f"model.{idx}"
            if kind == "stem":
                self.mod[idx] = dict(kind="conv", blk=_Block(self, pre, 8, p["cout"], 3, 2, cin_real=3))
            elif kind == "conv":
                self.mod[idx] = dict(kind="conv", blk=_Block(self, pre, p["cin"], p["cout"], 3, 2))
            elif kind == "c2f":
                c = p["cout"] // 2
                m = dict(kind="c2f", c=c, n=p["n"], add=p["add"], p=p,
                         cv1=_Block(self, pre + ".cv1", p["cin"], 2 * c, 1, 1),
                         cv2=_Block(self, pre + ".cv2", (2 + p["n"]) * c, p["cout"], 1, 1), m=[])
                for j in range(p["n"]):
                    m["m"].append((_Block(self, pre + f".m.{j}.cv1", c, c, 3, 1), _Block(self, pre + f".m.{j}.cv2", c, c, 3, 1)))
                self.mod[idx] = m
            elif kind == "sppf":
                self.mod[idx] = dict(kind="sppf", c_=p["cin"] // 2, cv1=_Block(self, pre + ".cv1", p["cin"], p["cin"] // 2, 1, 1),
                                     cv2=_Block(self, pre + ".cv2", p["cin"] * 2, p["cout"], 1, 1))
        ch = (_c(256, sc), _c(512, sc), _c(1024, sc))
        self.c2 = max(16, ch[0] // 4, REG_MAX * 4)
        self.c3 = max(ch[0], min(nc, 100))
        self.det = []
        for s, ci in enumerate(ch):
            pre = f"model.22"
            self.det.append(dict(
                b0=_Block(self, f"{pre}.cv2.{s}.0", ci, self.c2, 3, 1), b1=_Block(self, f"{pre}.cv2.{s}.1", self.c2, self.c2, 3, 1),
                b2=_Block(self, f"{pre}.cv2.{s}.2", self.c2, 4 * REG_MAX, 1, 1, bn=False),
                c0=_Block(self, f"{pre}.cv3.{s}.0", ci, self.c3, 3, 1), c1=_Block(self, f"{pre}.cv3.{s}.1", self.c3, self.c3, 3, 1),
                c2=_Block(self, f"{pre}.cv3.{s}.2", self.c3, self.ncp, 1, 1, bn=False, cout_real=nc)))

    def _alloc_buffers(self):
        B, S, dev = self.B, self.size, self.dev
        A = lambda h, c: _Act(self, B, h, h, c)
        self.x0 = A(S, 8)
        self.out: Dict[int, _Act] = {}
        self.aux: Dict[int, dict] = {}
        for idx, kind, p in self.layers:
            h = S // LAYER_STRIDE[idx]
            self.out[idx] = A(h, p["cout"])
            if kind == "c2f":
                c, n = p["cout"] // 2, p["n"]
                a = dict(y=A(h, (2 + n) * c), t=[A(h, c) for _ in range(n)])
                if "a" in p:
                    a["cat"] = A(h, p["cin"])
                self.aux[idx] = a
            elif kind == "sppf":
                self.aux[idx] = dict(y=A(h, p["cin"] * 2))
        self.det_act = []
        self.det_out = []
        for s, st in enumerate((8, 16, 32)):
            h = S // st
            self.det_act.append(dict(b0=A(h, self.c2), b1=A(h, self.c2), c0=A(h, self.c3), c1=A(h, self.c3)))
            T = B * h * h
            self.det_out.append(dict(box=torch.zeros((T, 4 * REG_MAX), device=dev), cls=torch.zeros((T, self.ncp), device=dev),
                                     dbox=torch.zeros((T, 4 * REG_MAX), device=dev), dcls=torch.zeros((T, self.ncp), device=dev)))
        # one allocation for every activation gradient (zeroed once per step)
        total = sum(a.buf.numel() for a in self._grad_specs)
        self.grad_flat = torch.zeros(total, dtype=torch.bfloat16, device=dev)
        pos = 0
        for a in self._grad_specs:
            n = a.buf.numel()
            a.grad = self.grad_flat[pos:pos + n].view(a.buf.shape)
            pos += n
        # per-block pre-activation, its gradient and the batch statistics
        self.z: Dict[str, torch.Tensor] = {}
        self.dz: Dict[str, torch.Tensor] = {}
        self.mean: Dict[str, torch.Tensor] = {}
        self.rstd: Dict[str, torch.Tensor] = {}
        self.geom: Dict[str, Tuple[int, int]] = {}
        ws_f, wd_n, col_n, zi_n, xp_n, dzp_n = 0, 0, 0, 0, 0, 0
        for b, (hin, hout) in self._block_geometry():
            T = B * hout * hout
            self.geom[b.key] = (hin, hout)
            if b.bn:
                self.z[b.key] = torch.zeros((_r64(T), b.cout), dtype=torch.bfloat16, device=dev)
                self.mean[b.key] = torch.zeros(b.cout, device=dev)
                self.rstd[b.key] = torch.zeros(b.cout, device=dev)
                ws_f = max(ws_f, bn_ws_floats(T, b.cout))
            else:
                ws_f = max(ws_f, colsum_ws_floats(T, b.cout))
            self.dz[b.key] = torch.zeros((_r64(T), b.cout), dtype=torch.bfloat16, device=dev)
            wd_n = max(wd_n, b.cout * b.taps * b.cin)
            if b.k == 3 and b.s == 1 and self.implicit_wgrad:
                hp = hin + 2                                   # operands of yv_wgrad_conv3 live on the zero-padded grid
                tpp = _r64(B * hp * hp)
                xp_n = max(xp_n, (tpp + 2 * (hp + 1)) * b.cin)
                dzp_n = max(dzp_n, tpp * b.cout)
            elif b.k == 3:
                col_n = max(col_n, _r64(T) * 9 * b.cin)
            if b.s == 2:
                zi_n = max(zi_n, B * hin * hin * b.cout)
        self.ws = torch.zeros(max(ws_f, 16), device=dev)
        self.wd_buf = torch.zeros(max(wd_n, 8), dtype=torch.bfloat16, device=dev)
        self.col = torch.zeros(max(col_n, 8), dtype=torch.bfloat16, device=dev)
        self.zi = torch.zeros(max(zi_n, 8), dtype=torch.bfloat16, device=dev)
        self.xp = torch.zeros(max(xp_n, 8), dtype=torch.bfloat16, device=dev)       # zero-initialised: its margins are read
        self.dzp = torch.zeros(max(dzp_n, 8), dtype=torch.bfloat16, device=dev)

    def _block_geometry(self):
        S = self.size
        for idx, kind, p in self.layers:
            h = S // LAYER_STRIDE[idx]
            m = self.mod[idx]
            if m["kind"] == "conv":
                yield m["blk"], (h * 2, h)
            elif m["kind"] == "c2f":
                yield m["cv1"], (h, h)
                yield m["cv2"], (h, h)
                for b1, b2 in m["m"]:
                    yield b1, (h, h)
                    yield b2, (h, h)
            else:
                yield m["cv1"], (h, h)
                yield m["cv2"], (h, h)
        for s, st in enumerate((8, 16, 32)):
            h = S // st
            for b in self.det[s].values():
                yield b, (h, h)

    # ------------------------------------------------------------------ one block, forward / backward
    def _fwd(self, b: _Block, x, out, res=None):
        """x: input view (B,Hin,Hin,cin); out / res: views on the output grid."""
        hin, hout = self.geom[b.key]
        T = self.B * hout * hout
        z = mview(self.z[b.key])
        conv_view(x, self.B, hout, hout, b.k, b.s, self.w16(b), b.cout, z)
        bn_stats(z, T, self.mean[b.key], self.rstd[b.key], self.run_mean[b.key], self.run_var[b.key], self.ws, BN_EPS, BN_MOMENTUM)
        bn_act_fwd(z, T, self.mean[b.key], self.rstd[b.key], self.p(b.gamma), self.p(b.beta), out, res=res)

    def _bwd(self, b: _Block, da, x_buf: torch.Tensor, x_off: int, dx=None):
        """da: gradient view of the block's output; x_buf[:, x_off:x_off+cin]: its input rows; dx: gradient view of
        the input (accumulated) or None."""
        hin, hout = self.geom[b.key]
        T = self.B * hout * hout
        Tp = _r64(T)
        dz = self.dz[b.key]
        if b.bn:
            bn_act_bwd(da, mview(self.z[b.key]), T, self.mean[b.key], self.rstd[b.key], self.p(b.gamma), self.p(b.beta),
                       self.gr(b.gamma), self.gr(b.beta), mview(dz), self.ws)
        else:                                                   # da is the f32 loss gradient (T, cout): cast + bias gradient
            cast_colsum(da, dz, self.gr(b.bias), self.ws)
        if self.overlap_wgrad:
            self._pending.append((b, x_buf, x_off))             # weight gradients run on the side stream (_flush_wgrads)
        else:
            self._wgrad_block(b, x_buf, x_off)
        if dx is not None:
            wd = self.wd_buf[:b.cin * b.taps * b.cout]
            conv_weight_dgrad(self.w16(b), b.cout, b.taps, b.cin, wd)
            if b.s == 1:
                src = mview(dz)
            else:
                zi = self.zi[:self.B * hin * hin * b.cout].view(self.B * hin * hin, b.cout)
                view_op(VIEW_ZERO_INSERT, mview(dz), mview(zi), self.B, hout, hout)
                src = mview(zi)
            conv_view(src, self.B, hin, hin, b.k, 1, wd.view(b.cin, b.taps * b.cout), b.cin, dx, res=dx)

    def _wgrad_block(self, b: _Block, x_buf: torch.Tensor, x_off: int):
        hin, hout = self.geom[b.key]
        T = self.B * hout * hout
        Tp = _r64(T)
        dz = self.dz[b.key]
        dw = self.gr(b.w).view(b.cout, b.taps * b.cin)
        if b.k == 1:
            wgrad(dz, x_buf[:, x_off:x_off + b.cin], dw, T=Tp)
        elif b.s == 1 and self.implicit_wgrad:
            # no im2col: both operands are copied once onto the zero-padded pixel grid, where every tap is a constant row
            # offset and the three taps of a kernel row are contiguous (yv_wgrad_conv3)
            hp = hin + 2
            tpad = self.B * hp * hp
            tpp, mg = _r64(tpad), hp + 1
            xp = self.xp[mg * b.cin:(mg + tpp) * b.cin].view(tpp, b.cin)
            view_op(VIEW_PAD, mview(x_buf, x_off, b.cin), mview(xp), self.B, hin, hin)
            dzp = self.dzp[:tpp * b.cout].view(tpp, b.cout)
            view_op(VIEW_PAD, mview(dz), mview(dzp), self.B, hout, hout)
            if tpp != tpad:
                dzp[tpad:].zero_()
            wgrad_conv3(dzp, xp, dw, tpp, hp)
        else:
            col = self.col[:Tp * 9 * b.cin].view(Tp, 9 * b.cin)
            if Tp != T:
                col[T:].zero_()
            im2col3(mview(x_buf, x_off, b.cin), self.B, hin, hin, b.s, col)
            wgrad(dz, col, dw, T=Tp)

    def _flush_wgrads(self):
        """The weight gradients of the blocks back-propagated since the last flush go to a second HIP stream: nothing on
        the data-gradient chain depends on them (only the optimiser does), and the chain's kernels leave CUs idle.  Hazards:
        dz of a block is written once per step (main stream, before the event) and activations are read-only in backward;
        the im2col scratch is private to the side stream; gradient slices are disjoint."""
        if not self._pending:
            return
        if self.s_w is None:
            self.s_w = torch.cuda.Stream()
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.s_w):
            self.s_w.wait_event(ev)
            for b, x_buf, x_off in self._pending:
                self._wgrad_block(b, x_buf, x_off)
        self._pending = []

    # ------------------------------------------------------------------ forward
    def forward(self, images: torch.Tensor):
        """images (B,S,S,3) u8 on the device -> per scale (box logits (B*h*h, 64) f32, class logits (B*h*h, ncp) f32)."""
        B, S = self.B, self.size
        if images.dtype != torch.uint8 or tuple(images.shape) != (B, S, S, 3) or not images.is_cuda:
            raise YvError(f"images must be ({B},{S},{S},3) uint8 on the device")
        blob_nhwc8(images.contiguous(), self.x0.buf)
        o = self.out
        for idx, kind, p in self.layers:
            m = self.mod[idx]
            if m["kind"] == "conv":
                src = self.x0 if idx == 0 else o[idx - 1]
                self._fwd(m["blk"], src.v(), o[idx].v())
            elif m["kind"] == "c2f":
                a = self.aux[idx]
                if "a" in p:
                    (ia, ua), (ib, _) = p["a"], p["b"]
                    cat, ca = a["cat"], o[ia].C
                    if ua:
                        view_op(VIEW_UP2, o[ia].v(), cat.v(0, ca), B, o[ia].H, o[ia].W)
                    else:
                        view_op(VIEW_COPY, o[ia].v(), cat.v(0, ca), B, cat.H, cat.W)
                    view_op(VIEW_COPY, o[ib].v(), cat.v(ca, o[ib].C), B, cat.H, cat.W)
                    xin = cat
                else:
                    xin = o[idx - 1]
                y, c = a["y"], m["c"]
                self._fwd(m["cv1"], xin.v(), y.v(0, 2 * c))
                for j, (b1, b2) in enumerate(m["m"]):
                    src = (1 + j) * c
                    self._fwd(b1, y.v(src, c), a["t"][j].v())
                    self._fwd(b2, a["t"][j].v(), y.v(src + c, c), res=y.v(src, c) if m["add"] else None)
                self._fwd(m["cv2"], y.v(), o[idx].v())
            else:
                y, c_ = self.aux[idx]["y"], m["c_"]
                self._fwd(m["cv1"], o[idx - 1].v(), y.v(0, c_))
                sppf_pool(y.buf[:y.T].view(B, y.H, y.W, y.C), c_)
                self._fwd(m["cv2"], y.v(), o[idx].v())
        res = []
        for s, fidx in enumerate((15, 18, 21)):
            f, d, act, do = o[fidx], self.det[s], self.det_act[s], self.det_out[s]
            h = f.H
            self._fwd(d["b0"], f.v(), act["b0"].v())
            self._fwd(d["b1"], act["b0"].v(), act["b1"].v())
            conv_view(act["b1"].v(), B, h, h, 1, 1, self.w16(d["b2"]), 4 * REG_MAX, mview(do["box"]), bias=self.p(d["b2"].bias),
                      out_f32=True)
            self._fwd(d["c0"], f.v(), act["c0"].v())
            self._fwd(d["c1"], act["c0"].v(), act["c1"].v())
            conv_view(act["c1"].v(), B, h, h, 1, 1, self.w16(d["c2"]), self.ncp, mview(do["cls"]), bias=self.p(d["c2"].bias),
                      out_f32=True)
            res.append((do["box"], do["cls"]))
        return res

    # ------------------------------------------------------------------ backward
    def backward(self, dlogits=None):
        """dlogits: per scale (d box (T,64) f32, d cls (T,ncp) f32); default: the buffers the loss kernel filled."""
        B = self.B
        self.grad_flat.zero_()
        o = self.out
        for s, fidx in enumerate((15, 18, 21)):
            f, d, act, do = o[fidx], self.det[s], self.det_act[s], self.det_out[s]
            dbox, dcls = (do["dbox"], do["dcls"]) if dlogits is None else dlogits[s]
            self._bwd(d["b2"], dbox, act["b1"].buf, 0, act["b1"].g())
            self._bwd(d["b1"], act["b1"].g(), act["b0"].buf, 0, act["b0"].g())
            self._bwd(d["b0"], act["b0"].g(), f.buf, 0, f.g())
            self._bwd(d["c2"], dcls, act["c1"].buf, 0, act["c1"].g())
            self._bwd(d["c1"], act["c1"].g(), act["c0"].buf, 0, act["c0"].g())
            self._bwd(d["c0"], act["c0"].g(), f.buf, 0, f.g())
            self._flush_wgrads()
        for idx, kind, p in reversed(self.layers):
            m = self.mod[idx]
            if m["kind"] == "conv":
                src = self.x0 if idx == 0 else o[idx - 1]
                self._bwd(m["blk"], o[idx].g(), src.buf, 0, None if idx == 0 else src.g())
            elif m["kind"] == "c2f":
                a = self.aux[idx]
                y, c = a["y"], m["c"]
                xin = a["cat"] if "a" in p else o[idx - 1]
                self._bwd(m["cv2"], o[idx].g(), y.buf, 0, y.g())
                for j in range(m["n"] - 1, -1, -1):
                    b1, b2 = m["m"][j]
                    src = (1 + j) * c
                    t = a["t"][j]
                    self._bwd(b2, y.g(src + c, c), t.buf, 0, t.g())
                    if m["add"]:
                        view_op(VIEW_ADD, y.g(src + c, c), y.g(src, c), B, y.H, y.W)
                    self._bwd(b1, t.g(), y.buf, src, y.g(src, c))
                self._bwd(m["cv1"], y.g(0, 2 * c), xin.buf, 0, xin.g())
                if "a" in p:
                    (ia, ua), (ib, _) = p["a"], p["b"]
                    cat, ca = a["cat"], o[ia].C
                    if ua:
                        view_op(VIEW_UP2_BWD, cat.g(0, ca), o[ia].g(), B, o[ia].H, o[ia].W)
                    else:
                        view_op(VIEW_ADD, cat.g(0, ca), o[ia].g(), B, cat.H, cat.W)
                    view_op(VIEW_ADD, cat.g(ca, o[ib].C), o[ib].g(), B, cat.H, cat.W)
            else:
                y, c_ = self.aux[idx]["y"], m["c_"]
                self._bwd(m["cv2"], o[idx].g(), y.buf, 0, y.g())
                for q in (2, 1, 0):                       # p_{q+1} = maxpool(p_q)
                    maxpool5_bwd(y.v(q * c_, c_), y.g((q + 1) * c_, c_), y.g(q * c_, c_), B, y.H, y.W)
                self._bwd(m["cv1"], y.g(0, c_), o[idx - 1].buf, 0, o[idx - 1].g())
            self._flush_wgrads()
        if self.s_w is not None:
            torch.cuda.current_stream().wait_stream(self.s_w)      # every weight gradient is in G before the caller goes on

    # ------------------------------------------------------------------ loss / step
    def loss(self, gt_boxes: torch.Tensor, gt_labels: torch.Tensor, gt_counts: torch.Tensor, gains=(7.5, 0.5, 1.5)):
        """v8 detection loss of the last forward; fills the d-logit buffers backward() consumes.
        gt_boxes (B,G,4) f32 xyxy input pixels, gt_labels (B,G) i32, gt_counts (B) i32, all on the device.
        Returns the device tensor {total*B, box, cls, dfl} (no host sync)."""
        B, G = self.B, gt_boxes.shape[1]
        if tuple(gt_boxes.shape) != (B, G, 4) or tuple(gt_labels.shape) != (B, G) or tuple(gt_counts.shape) != (B,):
            raise YvError("targets must be gt_boxes (B,G,4), gt_labels (B,G), gt_counts (B)")
        A = sum((self.size // st) ** 2 for st in (8, 16, 32))
        if G not in self._loss_ws:
            self._loss_ws[G] = torch.zeros(detect_loss_ws_bytes(B, A, G), dtype=torch.uint8, device=self.dev)
        do = self.det_out
        detect_loss([d["box"] for d in do], [d["cls"] for d in do], [d["dbox"] for d in do], [d["dcls"] for d in do], B,
                    self.size, self.nc, self.ncp, gt_boxes, gt_labels, gt_counts, self.loss_out, self._loss_ws[G], gains)
        return self.loss_out

    def step(self, images: torch.Tensor, gt_boxes: torch.Tensor, gt_labels: torch.Tensor, gt_counts: torch.Tensor,
             lr: Optional[float] = None, accumulate: int = 1, lrs: Optional[Dict[str, float]] = None,
             momentum: Optional[float] = None):
        """forward -> loss -> backward -> [every `accumulate` calls: data-parallel SUM all-reduce (mean folded into the
        update) -> optimiser -> EMA].  `lrs` gives per-group learning rates ({'w','bnw','bias'}: warm-up treats biases
        differently), `momentum` overrides momentum / beta1 for this step."""
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.forward(images)
        loss = self.loss(gt_boxes, gt_labels, gt_counts)
        self.backward()
        if accumulate > 1:
            if self.G_acc is None:
                self.G_acc = torch.zeros_like(self.G)
            if self.accumulated == 0:
                self.G_acc.copy_(self.G)
            else:
                axpby(self.G_acc, self.G, 1.0, 1.0)
            self.accumulated += 1
            if self.accumulated < accumulate:
                return loss
            self.G.copy_(self.G_acc)
            self.accumulated = 0
        self.reducer.reset()
        self.reducer.finish()
        self.optimizer_step(lr, grad_scale=1.0 / world, lrs=lrs, momentum=momentum)
        return loss

    # ------------------------------------------------------------------ optimiser
    def optimizer_step(self, lr: Optional[float] = None, grad_scale: float = 1.0, lrs: Optional[Dict[str, float]] = None,
                       momentum: Optional[float] = None):
        """One update of every parameter group (ultralytics: weight decay on conv weights only; BatchNorm weights and
        all biases undecayed) with torch.optim.SGD / SGD(nesterov) / AdamW semantics, then the ModelEMA update."""
        lr = self.lr if lr is None else lr
        mom = self.momentum if momentum is None else momentum
        self.step_count += 1
        t = self.step_count
        for g, wd in (("w", self.weight_decay), ("bnw", 0.0), ("bias", 0.0)):
            lo, hi = self.group[g]
            if hi <= lo:
                continue
            glr = lrs[g] if lrs is not None and g in lrs else lr
            mirror = self.P16 if g == "w" else None
            if self.optimizer == "sgd":
                sgd_step(self.P[lo:hi], self.G[lo:hi], self.Mo[lo:hi], glr, mom, wd, t == 1, grad_scale, mirror=mirror)
            else:
                if self.optimizer == "adamw" and self.V is None:
                    self.V = torch.zeros_like(self.P)
                optim_step(OPT_ADAMW if self.optimizer == "adamw" else OPT_SGD_NESTEROV, self.P[lo:hi], self.G[lo:hi],
                           self.Mo[lo:hi], self.V[lo:hi] if self.V is not None else None, glr, t, beta1=mom, weight_decay=wd,
                           grad_scale=grad_scale, mirror=mirror)
        if self.use_ema:
            self.ema_updates += 1
            d = self.ema_decay * (1.0 - math.exp(-self.ema_updates / self.ema_tau))
            ema_update(self.P_ema, self.P, d)
            ema_update(self.RS_ema, self.RS, d)
