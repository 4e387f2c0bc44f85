#!/usr/bin/env python3
"""Dev tool (GPU box): discover the operand / scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 with yv_mx_probe."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch
import yvhip
dev = "cuda:0"
ONE = 0x38                                                    # e4m3 1.0
def run(a, b, sa, sb, opsel=0):
    d = torch.zeros(64, 4, device=dev)
    yvhip.check(yvhip.lib.yv_mx_probe(a.data_ptr(), b.data_ptr(), sa.data_ptr(), sb.data_ptr(), opsel, d.data_ptr(), None), "probe")
    torch.cuda.synchronize()
    return d.cpu()
ones = lambda: torch.full((64, 32), ONE, dtype=torch.uint8, device=dev)
zeros = lambda: torch.zeros((64, 32), dtype=torch.uint8, device=dev)
s127 = lambda: torch.full((64,), 127, dtype=torch.int32, device=dev)
# 1. all ones: every output 128
d = run(ones(), ones(), s127(), s127())
print("all ones:", d.unique().tolist())
# 2. which output row/col does lane l of the FIRST operand feed?
for l0 in (0, 5, 16, 37, 63):
    a = zeros(); a[l0] = ONE
    d = run(a, ones(), s127(), s127())
    nz = d.nonzero()
    print(f"first operand lane {l0}: nonzero outputs lanes {sorted(set(nz[:,0].tolist()))[:20]} regs {sorted(set(nz[:,1].tolist()))} value {d[d!=0].unique().tolist()}")
for l0 in (0, 5, 16, 37):
    b = zeros(); b[l0] = ONE
    d = run(ones(), b, s127(), s127())
    nz = d.nonzero()
    print(f"second operand lane {l0}: nonzero outputs lanes {sorted(set(nz[:,0].tolist()))[:20]} regs {sorted(set(nz[:,1].tolist()))}")
# 3. K pairing: a single byte in a, a single byte in b (same row pair 0/0): which (lane group, byte) of b meets a's?
for (ga, ta) in ((0, 0), (0, 17), (1, 3), (2, 31), (3, 16)):
    a = zeros(); a[ga * 16 + 0, ta] = ONE
    hits = []
    for gb in range(4):
        for tb in range(32):
            b = zeros(); b[gb * 16 + 0, tb] = ONE
            d = run(a, b, s127(), s127())
            if float(d.abs().sum()) != 0:
                hits.append((gb, tb))
    print(f"a (group {ga}, byte {ta}) pairs with b {hits}")
# 4. whose scale register applies to the data of lane group g0 (row 0)?  a = ones only in (row 0, group g0)
for g0 in range(4):
    a = zeros(); a[g0 * 16 + 0] = ONE
    base = run(a, ones(), s127(), s127())[0, 0].item()
    eff = []
    for l in range(64):
        sa = s127(); sa[l] = 128
        v = run(a, ones(), sa, s127())
        if float((v - run(a, ones(), s127(), s127())).abs().sum()) != 0:
            eff.append(l)
    print(f"data in first-operand lane {g0*16} (group {g0}): base {base}, scale registers that change the result: {eff}")
# 5. opsel: scale byte k of the register
import numpy as np
sa = torch.from_numpy(np.full((64,), 127 | (128 << 8) | (129 << 16) | (130 << 24), dtype=np.uint32).view(np.int32)).to(dev)
for op in range(4):
    d = run(ones(), ones(), sa, s127(), op)
    print("opsel", op, "->", d.unique().tolist())
