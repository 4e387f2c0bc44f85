"""GPU parity: NMS family, postprocess, crop gather, decode, loss, SGD vs the oracle
and the golden fixtures.  Bit-exact for indices / integer coordinates / f32 gathers."""
import random

import numpy as np
import pytest
import torch

from inputs import coord_image, seeded_boxes
from oracle import boxes as ob
from oracle import train as ot
from oracle import yolo as oy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def yv():
    import yvhip
    yvhip.require_gpu()
    assert yvhip.lib.yv_device_is_gfx950() == 1
    return yvhip


DEV = "cuda:0"


# ------------------------------------------------------------------ custom_nms
def test_custom_nms_golden(yv, golden):
    for c in golden["G7_custom_nms"]:
        if "explicit_boxes" in c:
            b, s = torch.tensor(c["explicit_boxes"]), torch.tensor(c["scores"])
        elif c["n"] == 0:
            b, s = torch.zeros(0, 4), torch.zeros(0)
        else:
            b, s = seeded_boxes(c["n"], c["seed"])
        assert yv.custom_nms(b, s, c["thr"]) == c["keep"], (c.get("n"), c["thr"])


def test_custom_nms_ties_and_degenerate(yv):
    # exact score ties -> (score desc, index asc), identical to the oracle's stable sort
    g = torch.Generator().manual_seed(1)
    b, _ = seeded_boxes(300, 21)
    s = torch.randint(0, 8, (300,), generator=g).float() / 8
    assert yv.custom_nms(b, s, 0.45) == ob.custom_nms(b, s, 0.45)
    # zero-area boxes: 0/0 IoU is NaN -> never "< thr" -> suppressed, like the reference mask
    b = torch.tensor([[1., 1., 1., 1.], [1., 1., 1., 1.], [0., 0., 4., 4.]])
    s = torch.tensor([.9, .8, .7])
    assert yv.custom_nms(b, s, 0.45) == ob.custom_nms(b, s, 0.45) == [0, 2]
    # negative zero / negative scores ordering
    s = torch.tensor([-0.0, 0.0, -1.0])
    b = torch.tensor([[0., 0., 1., 1.], [10., 10., 11., 11.], [20., 20., 21., 21.]])
    assert yv.custom_nms(b, s, 0.45) == ob.custom_nms(b, s, 0.45) == [0, 1, 2]


def test_custom_nms_batched_ragged(yv):
    S, n = 37, 100
    bs, ss, counts = [], [], []
    rng = random.Random(3)
    for i in range(S):
        b, s = seeded_boxes(n, 100 + i)
        bs.append(b); ss.append(s); counts.append(rng.choice([0, 1, 2, 17, 64, 65, 100]))
    B = torch.stack(bs).to(DEV); Sc = torch.stack(ss).to(DEV)
    cnt = torch.tensor(counts, dtype=torch.int32, device=DEV)
    keep, num = yv.custom_nms_batched(B, Sc, cnt, 0.45)
    keep, num = keep.cpu(), num.cpu()
    for i in range(S):
        exp = ob.custom_nms(bs[i][:counts[i]], ss[i][:counts[i]], 0.45)
        assert int(num[i]) == len(exp) and keep[i, :len(exp)].tolist() == exp
        assert torch.all(keep[i, len(exp):] == -1)


def test_custom_nms_full_size_properties(yv):
    # BASELINE-size property checks: idempotence, sortedness, mutual IoU < thr (n = 8400 raw anchors)
    b, s = seeded_boxes(8400, 99)
    keep = yv.custom_nms(b, s, 0.45)
    kb, ks = b[keep], s[keep]
    assert torch.all(ks[:-1] >= ks[1:])
    assert yv.custom_nms(kb, ks, 0.45) == list(range(len(keep)))
    iou = ob.box_iou(kb, kb)
    iou.fill_diagonal_(0)
    assert float(iou.max()) < 0.45


# ------------------------------------------------------------------ EfficientNMS contract
def _rand_dets(B, A, nc, seed, dense=False):
    g = torch.Generator().manual_seed(seed)
    boxes = torch.stack([seeded_boxes(A, seed * 131 + i)[0] for i in range(B)])
    if dense:
        scores = torch.rand(B, A, nc, generator=g)
    else:
        scores = torch.rand(B, A, nc, generator=g) ** 8
    return boxes, scores


@pytest.mark.parametrize("single_kernel", [False, True])
@pytest.mark.parametrize("B,A,nc,dense", [(3, 500, 5, False), (2, 8400, 5, False), (2, 8400, 5, True), (1, 64, 1, True),
                                          (2, 1000, 80, False), (5, 8400, 1, False), (3, 2100, 3, True)])
def test_efficient_nms_vs_oracle(yv, B, A, nc, dense, single_kernel):
    """Both device forms (multi-workgroup = the pipeline's, single workgroup per image) against oracle/boxes.py, bit-exact.
    Cases: sparse and dense score sets (dense: every score above the threshold -> the top-4096 selection path and classes
    with thousands of candidates), one class (everything in one greedy chain), 80 classes (mostly empty classes)."""
    boxes, scores = _rand_dets(B, A, nc, 5 + A + nc, dense)
    exp = ob.efficient_nms(boxes, scores)
    got = yv.efficient_nms(boxes.to(DEV), scores.to(DEV), single_kernel=single_kernel)
    for e, g in zip(exp, got):
        assert torch.equal(e, g.cpu())


@pytest.mark.parametrize("single_kernel", [False, True])
def test_efficient_nms_ties_on_cut_and_empty(yv, single_kernel):
    # many equal scores straddling the pre-NMS top-k cut: lowest flat index wins (defined behaviour)
    B, A, nc = 1, 8400, 5
    boxes, _ = _rand_dets(B, A, nc, 77)
    g = torch.Generator().manual_seed(9)
    scores = (torch.randint(0, 6, (B, A, nc), generator=g).float() / 8 + 0.3)
    exp = ob.efficient_nms(boxes, scores)
    got = yv.efficient_nms(boxes.to(DEV), scores.to(DEV), single_kernel=single_kernel)
    for e, g_ in zip(exp, got):
        assert torch.equal(e, g_.cpu())
    # nothing above threshold -> zero detections, zero padded
    got = yv.efficient_nms(boxes.to(DEV), torch.zeros(B, A, nc, device=DEV), single_kernel=single_kernel)
    assert int(got[0][0, 0]) == 0 and float(got[1].abs().sum()) == 0 and float(got[2].abs().sum()) == 0


def test_efficient_nms_suppression_chains(yv):
    """Worst case for the parallel-round tile resolution (nms.hip tile_resolve): chains in which candidate i overlaps only
    candidates i-1 and i+1, so whether i is kept depends on ALL earlier members (kept, dead, kept, ...: as many rounds as the
    tile has members), chains that straddle 64-wide tile boundaries, plus a single class with every box identical (one kept,
    hundreds suppressed across many tiles)."""
    g = torch.Generator().manual_seed(31)
    B, A, nc = 3, 400, 2
    x0 = torch.arange(A, dtype=torch.float32) * 7.0                   # width 20, step 7: IoU(i, i+1) = 13/27 = 0.48, IoU(i, i+2) = 6/34 = 0.18
    boxes = torch.stack([x0, torch.zeros(A), x0 + 20.0, torch.full((A,), 20.0)], -1)[None].repeat(B, 1, 1).contiguous()
    boxes[2] = torch.tensor([100.0, 100.0, 180.0, 190.0])            # image 2: identical boxes
    scores = torch.zeros(B, A, nc)
    scores[0, :, 0] = torch.linspace(0.95, 0.30, A)                   # image 0: one chain in rank order
    perm = torch.randperm(A, generator=g)
    scores[1, :, 0] = torch.linspace(0.95, 0.30, A)[perm]             # image 1: the same boxes, ranks shuffled along the chain
    scores[1, :, 1] = torch.linspace(0.90, 0.26, A)
    scores[2, :, 1] = torch.linspace(0.99, 0.31, A)
    for iou in (0.45, 0.15, 0.65):
        exp = ob.efficient_nms(boxes, scores, 0.25, iou, 100, 4096)
        for sk in (False, True):
            got = yv.efficient_nms(boxes.to(DEV), scores.to(DEV), 0.25, iou, 100, 4096, single_kernel=sk)
            for e, g_ in zip(exp, got):
                assert torch.equal(e, g_.cpu()), (iou, sk)
    assert int(exp[0][2, 0]) == 1 and int(exp[0][0, 0]) == 100


def test_efficient_nms_properties_at_bench_batch(yv):
    """Size-independent properties at the post-processing bench's full size (256 images x 8400 anchors x 5 classes, SURVEY 8(d)
    score / box distributions): the oracle checks the first 6 images bit for bit; for all 256 the output must be sorted by
    score, hold no pair of same-class boxes above the IoU threshold, carry the input's own scores, be zero padded - and be a
    fixed point: feeding the kept boxes back (one-hot class scores) keeps every one of them in the same order."""
    B, A, nc, thr, iou_thr, K = 256, 8400, 5, 0.25, 0.65, 100
    g = torch.Generator().manual_seed(4321)
    ctr = torch.rand(B, A, 2, generator=g) * 640
    wh = torch.rand(B, A, 2, generator=g) * 240 + 16
    boxes = torch.cat([(ctr - wh / 2).clamp(0, 640), (ctr + wh / 2).clamp(0, 640)], -1).contiguous()
    scores = torch.distributions.Beta(0.5, 4.0).sample((B, A, nc)).float()
    bd, sd = boxes.to(DEV), scores.to(DEV)
    num, kb, ks, kl = yv.efficient_nms(bd, sd, thr, iou_thr, K)
    exp = ob.efficient_nms(boxes[:6], scores[:6], thr, iou_thr, K)
    for e, g_ in zip(exp, (num[:6], kb[:6], ks[:6], kl[:6])):
        assert torch.equal(e, g_.cpu())
    n = num[:, 0].long()
    slot = torch.arange(K, device=DEV)[None]
    live = slot < n[:, None]
    assert bool((n <= K).all()) and int(n.min()) > 0
    assert float(ks[~live].abs().sum()) == 0 and float(kb[~live].abs().sum()) == 0 and int(kl[~live].abs().sum()) == 0
    d = ks[:, 1:] - ks[:, :-1]
    assert bool((d[live[:, 1:]] <= 0).all())                          # sorted by score
    assert bool((ks[live] > thr).all())
    # every kept (box, score, label) is one of the image's candidates: its score is the anchor's score for that class
    same_box = (kb[:, :, None, :] == bd[:, None, :, :]).all(-1)        # (B, K, A) - boolean, chunked to bound memory
    for b0 in range(0, B, 32):
        sb = same_box[b0:b0 + 32]
        cand = torch.gather(sd[b0:b0 + 32].permute(0, 2, 1), 1, kl[b0:b0 + 32].long()[:, :, None].expand(-1, -1, A))   # (b, K, A)
        hit = (sb & (cand == ks[b0:b0 + 32][:, :, None])).any(-1)
        assert bool(hit[live[b0:b0 + 32]].all())
    # no same-class pair above the threshold
    x1 = torch.maximum(kb[:, :, None, 0], kb[:, None, :, 0]); y1 = torch.maximum(kb[:, :, None, 1], kb[:, None, :, 1])
    x2 = torch.minimum(kb[:, :, None, 2], kb[:, None, :, 2]); y2 = torch.minimum(kb[:, :, None, 3], kb[:, None, :, 3])
    inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
    area = (kb[..., 2] - kb[..., 0]) * (kb[..., 3] - kb[..., 1])
    iou = inter / (area[:, :, None] + area[:, None, :] - inter).clamp(min=1e-9)
    pair = live[:, :, None] & live[:, None, :] & (kl[:, :, None] == kl[:, None, :]) & ~torch.eye(K, dtype=torch.bool, device=DEV)[None]
    assert float(iou[pair].max()) <= iou_thr + 1e-6
    # fixed point
    s2 = torch.zeros(B, K, nc, device=DEV)
    s2.scatter_(2, kl.long()[:, :, None], ks[:, :, None])
    s2[~live] = 0
    num2, kb2, ks2, kl2 = yv.efficient_nms(kb.contiguous(), s2, thr, iou_thr, K)
    assert torch.equal(num2, num) and torch.equal(kb2, kb) and torch.equal(ks2, ks) and torch.equal(kl2, kl)


def test_efficient_nms_clustered_boxes_and_small_limits(yv):
    """Heavy suppression (clusters of near-identical boxes: many tiles are walked before max_out boxes are kept, kept lists
    of several classes interleave in the merge), small max_out / pre_topk, threshold variations; the two device forms must
    also agree with each other on a batch of 64 images."""
    g = torch.Generator().manual_seed(123)
    B, A, nc = 4, 3000, 4
    centres = torch.rand(B, 40, 2, generator=g) * 560 + 40
    pick = torch.randint(0, 40, (B, A), generator=g)
    c = torch.gather(centres, 1, pick[..., None].expand(B, A, 2)) + torch.randn(B, A, 2, generator=g) * 3
    wh = 60 + torch.randn(B, A, 2, generator=g).abs() * 6
    boxes = torch.cat([c - wh / 2, c + wh / 2], -1).contiguous()
    scores = torch.rand(B, A, nc, generator=g) ** 3
    for thr, iou, mo, topk in ((0.25, 0.65, 100, 4096), (0.05, 0.5, 7, 300), (0.6, 0.3, 100, 4096), (0.0, 0.65, 100, 1000)):
        exp = ob.efficient_nms(boxes, scores, thr, iou, mo, topk)
        for sk in (False, True):
            got = yv.efficient_nms(boxes.to(DEV), scores.to(DEV), thr, iou, mo, topk, single_kernel=sk)
            for e, g_ in zip(exp, got):
                assert torch.equal(e, g_.cpu()), (thr, iou, mo, topk, sk)
    # the same clusters with 12 classes: the per-class / merge pair of launches (more than 8 classes) on a segmented candidate
    # list (24,000 scores per image), images the head cannot finish, classes without candidates
    nc2 = 12
    s12 = torch.rand(B, A, nc2, generator=g) ** 3
    s12[:, :, 7] = 0.0                                                 # an empty class
    for thr, iou, mo, topk in ((0.5, 0.5, 100, 4096), (0.2, 0.65, 100, 4096), (0.7, 0.3, 50, 600)):
        exp = ob.efficient_nms(boxes, s12, thr, iou, mo, topk)
        for sk in (False, True):
            got = yv.efficient_nms(boxes.to(DEV), s12.to(DEV), thr, iou, mo, topk, single_kernel=sk)
            for e, g_ in zip(exp, got):
                assert torch.equal(e, g_.cpu()), (thr, iou, mo, topk, sk)
    bb, ss = _rand_dets(64, 8400, 5, 4242)
    a = yv.efficient_nms(bb.to(DEV), ss.to(DEV))
    b = yv.efficient_nms(bb.to(DEV), ss.to(DEV), single_kernel=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert int(a[0].sum()) > 0


# ------------------------------------------------------------------ postprocess + compaction
def _oracle_post(num, bb, sc, lb, ratio, dwdh, wh, conf, thr, mode, cap):
    idx, ib, s, l = ob.restore_and_filter(num, bb, sc, lb, ratio, dwdh, conf, mode)
    n = int(num)
    d = torch.tensor([dwdh[0], dwdh[1], dwdh[0], dwdh[1]], dtype=torch.float32)
    fb = (bb[:n].float() - d) / torch.tensor(ratio, dtype=torch.float32)
    sel = torch.tensor(idx, dtype=torch.long)
    keep = ob.custom_nms(fb[sel], sc[:n][sel], thr) if len(idx) else []
    if cap > 0:
        keep = keep[:cap]
    dets = []
    for k in keep:
        x0, y0, x1, y1 = ib[k]
        rect = ob.inflate_eval(x0, y0, x1, y1, wh[0], wh[1])
        dets.append((ib[k], s[k], l[k], list(rect), int(rect[2] > rect[0] and rect[3] > rect[1])))
    return dets


@pytest.mark.parametrize("mode", ["trunc", "round"])
def test_postprocess_vs_oracle(yv, mode):
    B, K = 9, 100
    boxes, scores = _rand_dets(B, 2000, 5, 41)
    num, bb, sc, lb = ob.efficient_nms(boxes, scores)
    rng = random.Random(5)
    ratio = [1.0, 0.5, 0.3333333, 0.75, 1.0, 0.6, 0.9, 0.41, 1.0]
    dwdh = [(0.0, 0.0), (0.0, 80.0), (13.5, 0.0), (0.0, 40.0), (0.0, 0.0), (10.0, 0.0), (0.0, 3.5), (7.0, 9.0), (0., 0.)]
    wh = [(640, 640), (1280, 960), (1839, 1920), (853, 746), (640, 640), (1033, 1066), (711, 703), (1526, 1517), (64, 48)]
    num[4, 0] = 0                       # an image without detections
    out = yv.postprocess_dets(num.to(DEV), bb.to(DEV), sc.to(DEV), lb.to(DEV),
                              torch.tensor(ratio, device=DEV), torch.tensor(dwdh, device=DEV).reshape(-1),
                              torch.tensor(wh, dtype=torch.int32, device=DEV).reshape(-1),
                              conf=0.35, dedupe_iou=0.45, coord_mode=mode, max_crops=0)
    out = {k: v.cpu() for k, v in out.items()}
    total_exp = []
    for b in range(B):
        dets = _oracle_post(num[b, 0], bb[b], sc[b], lb[b], ratio[b], dwdh[b], wh[b], 0.35, 0.45, mode, 0)
        assert int(out["det_count"][b]) == len(dets), b
        for k, (ibox, s, l, rect, ok) in enumerate(dets):
            assert out["det_box"][b, k].tolist() == ibox
            assert float(out["det_score"][b, k]) == s and int(out["det_label"][b, k]) == l
            assert out["crop_rect"][b, k].tolist() == rect and int(out["crop_ok"][b, k]) == ok
            if ok:
                total_exp.append([b] + rect + [k])
    cl, tot = yv.compact_crops(out["det_count"].to(DEV), out["crop_rect"].to(DEV), out["crop_ok"].to(DEV), 1024)
    assert int(tot[0]) == len(total_exp)
    assert cl.cpu()[:len(total_exp)].tolist() == total_exp
    # capacity smaller than the total truncates in order
    cl2, tot2 = yv.compact_crops(out["det_count"].to(DEV), out["crop_rect"].to(DEV), out["crop_ok"].to(DEV), 5)
    assert int(tot2[0]) == 5 and cl2.cpu().tolist() == total_exp[:5]


def test_postprocess_inflate_golden(yv, golden):
    """crop_image's inflate (G1) through the device kernel: one detection per 'image'."""
    cases = [c for c in golden["G1_crop_eval"]]
    B = len(cases)
    bb = torch.zeros(B, 100, 4); sc = torch.zeros(B, 100); lb = torch.zeros(B, 100, dtype=torch.int32)
    num = torch.ones(B, 1, dtype=torch.int32)
    for i, c in enumerate(cases):
        bb[i, 0] = torch.tensor(c["box"], dtype=torch.float32)
        sc[i, 0] = 0.9
    wh = torch.tensor([[c["W"], c["H"]] for c in cases], dtype=torch.int32)
    out = yv.postprocess_dets(num.to(DEV), bb.to(DEV), sc.to(DEV), lb.to(DEV), torch.ones(B, device=DEV),
                              torch.zeros(2 * B, device=DEV), wh.reshape(-1).to(DEV))
    rect = out["crop_rect"].cpu(); ok = out["crop_ok"].cpu()
    for i, c in enumerate(cases):
        x0, y0, x1, y1 = rect[i, 0].tolist()
        if "error" in c or c.get("empty"):
            assert int(ok[i, 0]) == 0
        else:
            assert [x0, y0] == c["origin"] and [x1 - x0, y1 - y0] == c["size"] and int(ok[i, 0]) == 1


def test_postprocess_max_crops_and_no_dedupe(yv):
    boxes, scores = _rand_dets(2, 3000, 5, 8, dense=True)
    num, bb, sc, lb = ob.efficient_nms(boxes, scores)
    args = (num.to(DEV), bb.to(DEV), sc.to(DEV), lb.to(DEV), torch.ones(2, device=DEV), torch.zeros(4, device=DEV),
            torch.tensor([640, 640, 640, 640], dtype=torch.int32, device=DEV))
    o4 = yv.postprocess_dets(*args, max_crops=4)
    o0 = yv.postprocess_dets(*args, max_crops=0)
    assert o4["det_count"].tolist() == [4, 4]
    assert torch.equal(o4["det_box"][:, :4], o0["det_box"][:, :4])
    on = yv.postprocess_dets(*args, dedupe_iou=0.0)
    assert on["det_count"].tolist() == [int((sc[b, :int(num[b, 0])] >= 0.35).sum()) for b in range(2)]


# ------------------------------------------------------------------ crop gather
def test_crop_resize_norm_bit_exact(yv):
    W, H = 640, 480
    g = torch.Generator().manual_seed(12)
    imgs = torch.randint(0, 256, (3, H, W, 3), generator=g, dtype=torch.uint8)
    imgs[0] = torch.from_numpy(coord_image(W, H))
    rects = [(0, 78, 33, 298, 253, 0), (0, 0, 0, 38, 30, 1), (1, 598, 438, 640, 480, 0), (1, 5, 5, 16, 14, 1),
             (2, 0, 0, 640, 480, 0), (2, 100, 100, 101, 101, 1), (2, 17, 3, 241, 227, 2), (0, 300, 200, 301, 480, 2)]
    cl = torch.tensor(rects, dtype=torch.int32, device=DEV)
    tot = torch.tensor([len(rects)], dtype=torch.int32, device=DEV)
    dimg = imgs.to(DEV)
    f32 = yv.crop_resize_norm(dimg, cl, tot, len(rects), 224, 16, layout=0).cpu()
    b16 = yv.crop_resize_norm(dimg, cl, tot, len(rects), 224, 16, layout=1).cpu()
    pm = yv.crop_resize_norm(dimg, cl, tot, len(rects), 224, 16, layout=2).cpu()
    pm8 = yv.crop_resize_norm(dimg, cl, tot, len(rects), 224, 8, layout=2).cpu()
    for r, rc in enumerate(rects):
        exp = ob.crop_resize_normalize(imgs[rc[0]].numpy(), rc[1:5])
        assert np.array_equal(f32[r].numpy(), exp), r                        # f32: bit exact
        e16 = torch.from_numpy(exp).to(torch.bfloat16)
        assert torch.equal(b16[r], e16)                                      # bf16: RNE of the same f32
        assert torch.equal(pm[r * 196:(r + 1) * 196], torch.from_numpy(ob.patchify(exp, 16)).to(torch.bfloat16))
        assert torch.equal(pm8[r * 784:(r + 1) * 784], torch.from_numpy(ob.patchify(exp, 8)).to(torch.bfloat16))
    # rows beyond crop_total are untouched (device-side dynamic batch)
    tot1 = torch.tensor([2], dtype=torch.int32, device=DEV)
    part = yv.crop_resize_norm(dimg, cl, tot1, len(rects), 224, 16, layout=0).cpu()
    assert torch.equal(part[:2], f32[:2]) and float(part[2:].abs().sum()) == 0


def test_crop_wide_sources_both_gather_paths(yv):
    """Camera-sized sources (1080 x 1920, a row is not a multiple of 8 bytes).  The patch-major layout stages its source rows
    through a 16.5 KB LDS buffer in passes of as many rows as fit: 16 rows for crops up to 336 pixels wide, 8 up to 680, 4 up to
    1,370, 2 beyond (aligned 8-byte loads around arbitrary byte offsets, first / last pixel of the buffer included) - the widths
    below sit on both sides of every pass boundary; the planar layouts gather from global memory.  Bit for bit against the
    oracle, and the same bits whatever number of row groups a block walks (yv_crop_debug)."""
    H, W = 1080, 1921
    g = torch.Generator().manual_seed(5)
    imgs = torch.randint(0, 256, (2, H, W, 3), generator=g, dtype=torch.uint8)
    rects = [(0, 0, 0, 1921, 1080, 0), (1, 1200, 1000, 1921, 1080, 0), (0, 0, 0, 677, 5, 0), (1, 1, 1, 679, 400, 0),
             (1, 1244, 3, 1921, 1080, 0), (0, 1920, 1079, 1921, 1080, 0), (1, 1913, 1070, 1921, 1080, 0), (0, 0, 0, 1, 1, 0),
             (1, 333, 777, 1009, 1011, 0), (0, 5, 5, 683, 300, 0),
             (0, 3, 10, 339, 200, 0), (0, 3, 10, 340, 200, 0), (1, 7, 20, 687, 300, 0), (1, 7, 20, 688, 333, 0),
             (0, 1, 0, 1371, 500, 0), (0, 1, 0, 1372, 500, 0), (1, 0, 100, 1500, 1080, 0)]
    cl = torch.tensor(rects, dtype=torch.int32, device=DEV)
    tot = torch.tensor([len(rects)], dtype=torch.int32, device=DEV)
    dimg = imgs.to(DEV)
    f32 = yv.crop_resize_norm(dimg, cl, tot, len(rects), 224, 16, layout=0).cpu()
    pm = yv.crop_resize_norm(dimg, cl, tot, len(rects), 224, 16, layout=2).cpu()
    for r, rc in enumerate(rects):
        exp = ob.crop_resize_normalize(imgs[rc[0]].numpy(), rc[1:5])
        assert np.array_equal(f32[r].numpy(), exp), r
        assert torch.equal(pm[r * 196:(r + 1) * 196], torch.from_numpy(ob.patchify(exp, 16)).to(torch.bfloat16)), r
    try:
        for gpb in (1, 2, 5, 14):
            yv.lib.yv_crop_debug(gpb)
            assert torch.equal(yv.crop_resize_norm(dimg, cl, tot, len(rects), 224, 16, layout=2).cpu(), pm), gpb
    finally:
        yv.lib.yv_crop_debug(0)


def test_crop_gather_at_bench_batch(yv):
    """The crop gather at the post-processing bench's full size (256 images of 640 x 640, 1024 crops, patch-major bf16): 24
    crops sampled across the batch bit for bit against the oracle; for every crop the three layouts must hold the same
    values (the patch-major image un-patchified equals the planar bf16 image = RNE of the planar f32 image), and every value
    lies in [-1, 1] on the 256-level grid (x - 127.5) / 127.5."""
    B, S, R, P = 256, 640, 1024, 16
    g = torch.Generator().manual_seed(77)
    imgs = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8)
    img_idx = torch.arange(R) % B
    x0 = torch.randint(0, S - 8, (R,), generator=g); y0 = torch.randint(0, S - 8, (R,), generator=g)
    w = torch.randint(1, 400, (R,), generator=g); h = torch.randint(1, 400, (R,), generator=g)
    x1 = torch.minimum(x0 + w, torch.tensor(S)); y1 = torch.minimum(y0 + h, torch.tensor(S))
    rects = torch.stack([img_idx, x0, y0, x1, y1, torch.zeros(R, dtype=torch.long)], 1).to(torch.int32)
    cl = rects.to(DEV); tot = torch.tensor([R], dtype=torch.int32, device=DEV)
    dimg = imgs.to(DEV)
    pm = yv.crop_resize_norm(dimg, cl, tot, R, 224, P, layout=2)                    # (R * 196, 768) bf16
    b16 = yv.crop_resize_norm(dimg, cl, tot, R, 224, P, layout=1)                   # (R, 3, 224, 224) bf16
    f32 = yv.crop_resize_norm(dimg, cl, tot, R, 224, P, layout=0)                   # (R, 3, 224, 224) f32
    assert torch.equal(f32.to(torch.bfloat16), b16)
    unp = pm.view(R, 14, 14, 3, P, P).permute(0, 3, 1, 4, 2, 5).reshape(R, 3, 224, 224)
    assert torch.equal(unp, b16)
    lvl = f32 * 127.5 + 127.5
    assert float(f32.abs().max()) <= 1.0 and float((lvl - lvl.round()).abs().max()) < 1e-3
    for r in torch.randperm(R, generator=g)[:24].tolist():
        rc = rects[r].tolist()
        exp = ob.crop_resize_normalize(imgs[rc[0]].numpy(), rc[1:5])
        assert np.array_equal(f32[r].cpu().numpy(), exp), r


# ------------------------------------------------------------------ decode
def test_detect_decode_vs_oracle(yv):
    B, nc, size = 2, 5, 640
    g = torch.Generator().manual_seed(4)
    raw = torch.randn(B, 64 + nc, 8400, generator=g) * 2
    eb, es = oy.decode(raw, nc, size)
    box_l, cls_l, a0 = [], [], 0
    for s in (8, 16, 32):
        w = size // s
        part = raw[:, :, a0:a0 + w * w].reshape(B, 64 + nc, w, w).permute(0, 2, 3, 1).contiguous()
        box_l.append(part[..., :64].contiguous().to(DEV))
        c = torch.zeros(B, w, w, 8); c[..., :nc] = part[..., 64:]
        cls_l.append(c.to(DEV))
        a0 += w * w
    gb, gs = yv.detect_decode(box_l, cls_l, size, nc)
    # fp tolerance: exp() implementations differ; coordinates are O(100) px
    assert torch.allclose(gb.cpu(), eb, atol=2e-3, rtol=1e-5)
    assert torch.allclose(gs.cpu(), es, atol=1e-6, rtol=1e-5)


# ------------------------------------------------------------------ loss / SGD
def test_loss_golden(yv, golden):
    for c in golden["G3_loss"]:
        x = torch.tensor(c["x"], device=DEV); lab = torch.tensor(c["label"], dtype=torch.int32, device=DEV)
        loss, grad = yv.loss_fwd_bwd(x, lab)
        assert abs(float(loss[0]) - c["total"]) < 2e-6 * max(1.0, abs(c["total"]))      # f32 tolerance
        assert torch.allclose(grad.cpu(), torch.tensor(c["grad"]), atol=1e-7, rtol=2e-5)


def test_sgd_step_bit_exact(yv):
    g = torch.Generator().manual_seed(2)
    n = 1000003
    p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g) * 0.01
    pe, me = ot.sgd_step(p, gr, None, 1e-4)
    pd, md = p.to(DEV), torch.zeros(n, device=DEV)
    yv.sgd_step(pd, gr.to(DEV), md, 1e-4, first=True)
    assert torch.equal(pd.cpu(), pe) and torch.equal(md.cpu(), me)
    gr2 = torch.randn(n, generator=g) * 0.01
    pe2, me2 = ot.sgd_step(pe, gr2, me, 9.75528e-5)
    yv.sgd_step(pd, gr2.to(DEV), md, 9.75528e-5, first=False)
    assert torch.equal(pd.cpu(), pe2) and torch.equal(md.cpu(), me2)

@pytest.mark.gpu
@pytest.mark.parametrize("scale,nc,S,B", [("n", 5, 640, 3), ("s", 5, 320, 2), ("m", 3, 128, 2), ("n", 16, 96, 1)])
def test_fused_detect_tail_is_bit_identical(scale, nc, S, B):
    """yv_detect_tail (last 1 x 1 convolutions of both Detect branches + DFL decode + sigmoid, three scales in one launch) against
    the unfused sequence (two conv launches per scale into f32 logit buffers + yv_detect_decode): same accumulation order, same
    decode statements -> the same bits, for c3 = 64 / 128 / 192 (YOLOv8 n / s / m), ragged last 16-pixel groups and nc up to 16."""
    import yvhip
    from yvhip import engines
    yvhip.require_gpu()
    eng = engines.YoloEngine(engines.init_yolo_state(scale, nc, seed=3, head_gain=4.0), scale, nc, S, "cuda:0")
    assert eng.fused_tail
    g = torch.Generator().manual_seed(S + nc)
    img = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8).to("cuda:0")
    boxes_f, scores_f = eng(img)
    eng.fused_tail = False
    boxes_u, scores_u = eng(img)
    torch.cuda.synchronize()
    assert torch.equal(boxes_f, boxes_u)
    assert torch.equal(scores_f, scores_u)
    assert float(scores_f.max()) > 0.3 and float(boxes_f.abs().max()) > 10          # a non-trivial head
