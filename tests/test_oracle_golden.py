"""The oracle pinned against vectors captured from the reference's own
functions (tests/golden/make_golden.py).  CPU only."""
import math
import random

import numpy as np
import torch

from inputs import seeded_boxes
from oracle import boxes as ob
from oracle import train as ot
from oracle import vit as ov
from oracle import yolo as oy


def test_g1_crop_eval(golden):
    n_ok = 0
    for c in golden["G1_crop_eval"]:
        x0, y0, x1, y1 = ob.inflate_eval(*c["box"], c["W"], c["H"])
        if "error" in c:
            assert x1 < x0 or y1 < y0
        elif c.get("empty"):
            assert (x1 - x0, y1 - y0) == tuple(c["size"]) and (x1 == x0 or y1 == y0)
        else:
            assert [x0, y0] == c["origin"], c
            assert [x1 - x0, y1 - y0] == c["size"], c
            n_ok += 1
    assert n_ok > 80


def test_g2_crop_train(golden):
    for c in golden["G2_crop_train"]:
        rng = random.Random(c["seed"])
        x0, y0, x1, y1 = ob.inflate_train(*c["box"], c["W"], c["H"], rng)
        assert [x0, y0] == c["origin"] and [x1 - x0, y1 - y0] == c["size"], c


def test_g3_loss(golden):
    for c in golden["G3_loss"]:
        x = torch.tensor(c["x"], requires_grad=True)
        y = torch.nn.functional.one_hot(torch.tensor(c["label"]), 5).float()
        assert abs(float(ot.lsce(x, y)) - c["lsce"]) < 1e-6
        assert abs(float(ot.focal(x, y)) - c["focal"]) < 1e-6
        tot = ot.build_loss(x, y)
        assert abs(float(tot) - c["total"]) < 1e-6
        tot.backward()
        assert torch.allclose(x.grad, torch.tensor(c["grad"]), atol=1e-7, rtol=1e-5)
    # the known answer quoted in SURVEY.md section 8(a) C1
    c = golden["G3_loss"][1]
    assert abs(c["lsce"] - 1.984541893) < 1e-6 and abs(c["total"] - 0.734222889) < 1e-6


def test_g4_lr(golden):
    for E, tab in golden["G4_lr"].items():
        for t, v in enumerate(tab):
            assert ot.cosine_lr(t, int(E), 1e-4) == v


def test_g5_correct(golden):
    c = golden["G5_correct"]
    eq, cm = ot.get_correct(torch.tensor(c["out"]), torch.nn.functional.one_hot(torch.tensor(c["label"]), 5).float())
    assert eq.int().tolist() == c["eq"] and cm.tolist() == c["cm"]


def test_g6_wrapper(golden, golden_dir):
    z = np.load(golden_dir + "/G6_wrapper.npz")
    sd = {k.replace("__", "."): torch.from_numpy(z[k]) for k in z.files if k not in ("feats", "out")}
    assert sorted(sd) == sorted(golden["G6_keys"]) == ["fc.1.bias", "fc.1.weight", "fc.3.bias", "fc.3.weight"]
    out = ov.wrapper_head(sd, torch.from_numpy(z["feats"]))
    assert torch.allclose(out, torch.from_numpy(z["out"]), atol=1e-6, rtol=1e-6)


def test_g7_custom_nms(golden):
    for c in golden["G7_custom_nms"]:
        if "explicit_boxes" in c:
            b, s = torch.tensor(c["explicit_boxes"]), torch.tensor(c["scores"])
        elif c["n"] == 0:
            b, s = torch.zeros(0, 4), torch.zeros(0)
        else:
            b, s = seeded_boxes(c["n"], c["seed"])
        assert ob.custom_nms(b, s, c["thr"]) == c["keep"], (c.get("n"), c["thr"])


def test_structural_kats():
    # KAT-1 (test.ipynb:12) and the counts quoted in SURVEY.md section 8(a)
    assert oy.param_count("n", 5) == 3006623
    assert abs(2 * oy.macs("n", 5) / 1e9 - 8.1) < 0.05
    assert oy.param_count("s", 80) == 11156544 and oy.param_count("m", 80) == 25886080
    a, s = oy.make_anchors(640)
    assert a.shape == (8400, 2) and s.shape == (8400,)
    sd = ov.init_wrapper_state("vit_tiny_test")
    assert set(k for k in sd if not k.startswith("model.")) == {"fc.1.weight", "fc.1.bias", "fc.3.weight", "fc.3.bias"}


def test_vit_param_counts():
    # shapes only (no allocation of the big models' values beyond B/16)
    for name, n in (("vit_base_patch16_224", 86567656),):
        assert ov.backbone_param_count(ov.init_wrapper_state(name)) == n
    P, D, L, H = ov.vit_cfg("vit_large_patch16_224")
    n_l = D * 3 * P * P + D + D + 197 * D + L * (2 * D + 3 * D * D + 3 * D + D * D + D + 2 * D + 8 * D * D + 5 * D) + 2 * D + 1000 * D + 1000
    assert n_l == 304326632


def test_nearest_table_and_normalize():
    t = ob.nearest_index_table(224, 220)
    assert t[0] == 0 and t[-1] == 219 and len(t) == 224 and np.all(np.diff(t) >= 0)
    t = ob.nearest_index_table(224, 1000)
    assert t[-1] == int(math.floor(223 * (1000 / 224)))
    assert ob.nearest_index_table(224, 1).tolist() == [0] * 224
    x = ob.normalize_u8(np.array([0, 127, 128, 255], dtype=np.uint8))
    assert x[0] == -1.0 and x[3] == 1.0 and abs(x[1] + x[2]) < 1e-7


def test_efficient_nms_layout():
    b, s = seeded_boxes(500, 11)
    g = torch.Generator().manual_seed(3)
    sc = torch.rand(1, 500, 5, generator=g) * s[None, :, None]
    num, bb, ss, ll = ob.efficient_nms(b[None], sc)
    assert num.shape == (1, 1) and bb.shape == (1, 100, 4) and ss.shape == (1, 100) and ll.shape == (1, 100)
    assert num.dtype == torch.int32 and ll.dtype == torch.int32
    n = int(num[0, 0])
    assert 0 < n <= 100 and torch.all(ss[0, :n] > 0.25) and torch.all(ss[0, n:] == 0)
    assert torch.all(ss[0, :n - 1] >= ss[0, 1:n])
