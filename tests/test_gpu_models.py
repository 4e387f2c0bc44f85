"""GPU parity of the two network engines against the fp32 CPU oracle (parity UNPINNED for the
third-party math - see oracle/vit.py, oracle/yolo.py headers).  Tolerance:
bf16 path vs fp32 oracle - rel-L2 <= 2e-2 on logits / raw head outputs."""
import pytest
import torch

from oracle import boxes as ob
from oracle import vit as ov
from oracle import yolo as oy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("name,R", [("vit_tiny_test", 3), ("vit_base_patch16_224", 2), ("vit_tiny8_test", 2),
                                    ("vit_base_patch8_224", 1), ("vit_large_patch16_224", 1)])
def test_vit_engine_vs_oracle(name, R):
    from yvhip import engines
    sd = ov.init_wrapper_state(name, seed=11)
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(R, 3, 224, 224, generator=g) * 2 - 1).to(torch.bfloat16).float()
    ref_feats = ov.vit_forward(sd, x, name)
    ref_logits = ov.wrapper_head(sd, ref_feats)
    eng = engines.VitEngine(sd, name, 5)
    P = eng.P
    pm = torch.cat([torch.from_numpy(ob.patchify(x[r].numpy(), P)) for r in range(R)]).to(torch.bfloat16).to(DEV)
    cap = R + 1                                      # one spare slot: dynamic count leaves it untouched
    buf = eng.patch_buffer(cap)
    buf[:pm.shape[0]] = pm
    cnt = torch.tensor([R], dtype=torch.int32, device=DEV)
    feats = eng.backbone(buf, cap, cnt)
    logits = torch.zeros(cap, 5, device=DEV); labels = torch.full((cap,), -1, dtype=torch.int32, device=DEV)
    eng.head(feats, cap, logits, labels, count=cnt)
    torch.cuda.synchronize()
    assert rel_l2(feats[:R, :1000].cpu(), ref_feats) < 2e-2
    assert rel_l2(logits[:R].cpu(), ref_logits) < 2e-2
    assert float(feats[:R, 1000:].abs().sum()) == 0
    assert int(labels[R]) == -1 and float(logits[R].abs().sum()) == 0
    margin = ref_logits.topk(2, 1).values
    sure = (margin[:, 0] - margin[:, 1]) > 0.05 * ref_logits.abs().max()
    assert labels[:R].cpu()[sure].tolist() == ref_logits.argmax(1)[sure].tolist()


@pytest.mark.parametrize("scale,nc,size,B", [("n", 5, 128, 2), ("n", 5, 640, 1), ("s", 80, 64, 1), ("m", 80, 64, 1),
                                             ("s", 80, 640, 1)])
def test_yolo_engine_vs_oracle(scale, nc, size, B):
    from yvhip import engines
    sd = oy.init_state(scale, nc, seed=7)
    g = torch.Generator().manual_seed(2)
    img = torch.randint(0, 256, (B, size, size, 3), generator=g, dtype=torch.uint8)
    raw, outs = oy.forward_raw(sd, oy.blob(img), scale, nc, return_feats=True)
    eng = engines.YoloEngine(sd, scale, nc, size)
    box_l, cls_l = eng.forward_raw(img.to(DEV))
    torch.cuda.synchronize()
    bufs = eng._buffers(B)
    # intermediate feature maps (bf16 activations vs fp32 oracle)
    for idx in (0, 1, 2, 4, 6, 9, 12, 15, 18, 21):
        got = bufs["out"][idx].float().permute(0, 3, 1, 2).cpu()
        assert rel_l2(got, outs[idx]) < 2e-2, idx
    a0 = 0
    for s, st in enumerate((8, 16, 32)):
        w = size // st
        part = raw[:, :, a0:a0 + w * w].reshape(B, 64 + nc, w, w)
        assert rel_l2(box_l[s].permute(0, 3, 1, 2).cpu(), part[:, :64]) < 2e-2
        assert rel_l2(cls_l[s][..., :nc].permute(0, 3, 1, 2).cpu(), part[:, 64:]) < 2e-2
        a0 += w * w
    eb, es = oy.decode(raw, nc, size)
    gb, gs = eng(img.to(DEV))
    assert rel_l2(gs.cpu(), es) < 2e-2
    assert float((gb.cpu() - eb).abs().max()) < 0.02 * size       # pixels; DFL expectation of bf16 logits


def test_yolo_kat_param_count():
    from yvhip import engines
    tot = sum(ci * co * k * k + co for _, ci, co, k in engines.yolo_conv_keys("n", 5)) + 16
    assert tot == 3006623                                           # KAT-1, test.ipynb:12


def test_pipelined_runner_matches_single_stream():
    """Two-stream schedule (detector of batch i+1 overlapping classifier of batch i) and the split-classifier
    option must produce exactly the single-stream results for every batch in flight."""
    from yvhip import engines
    from yvhip.pipeline import DetectClassifyPipeline, PipelinedRunner
    name, S, B = "vit_tiny_test", 128, 4
    pipe = DetectClassifyPipeline(engines.YoloEngine(engines.init_yolo_state("n", 5, 3, 4.0), "n", 5, S, DEV),
                                  [engines.VitEngine(engines.init_vit_wrapper_state(name, 5, 4), name, 5, device=DEV)],
                                  max_crops_per_image=3)
    g = torch.Generator().manual_seed(11)
    batches = [torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8).to(DEV) for _ in range(5)]
    keys = ("det_count", "det_box", "det_score", "crop_list", "crop_total", "cls_logits", "cls_label")
    ref = []
    for im in batches:
        o = pipe(im)
        torch.cuda.synchronize()
        ref.append({k: o[k].clone() for k in keys})
    for split in (False, True):
        runner = PipelinedRunner(pipe, split_classifier=split)
        outs = [runner.submit(im) for im in batches]               # all five in flight back to back
        runner.sync()
        for o, r in zip(outs, ref):
            assert o["done"].query()
            for k in keys:
                assert torch.equal(o[k], r[k]), (split, k)


def test_pipelined_runner_with_dropped_results():
    """The throughput loop of bench.py: every batch's result dict is dropped as soon as the next batch is submitted,
    so the caching allocator may recycle the detector->classifier hand-off buffers while an earlier classifier pass is
    still queued.  Different images per batch; per-batch results are copied (on the classify stream) into buffers
    allocated up front and must equal the single-stream results."""
    from yvhip import engines
    from yvhip.pipeline import DetectClassifyPipeline, PipelinedRunner
    name, S, B = "vit_tiny_test", 128, 4
    pipe = DetectClassifyPipeline(engines.YoloEngine(engines.init_yolo_state("n", 5, 3, 4.0), "n", 5, S, DEV),
                                  [engines.VitEngine(engines.init_vit_wrapper_state(name, 5, 4), name, 5, device=DEV)],
                                  max_crops_per_image=3)
    g = torch.Generator().manual_seed(21)
    batches = [torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8).to(DEV) for _ in range(8)]
    keys = ("crop_list", "crop_total", "cls_logits", "cls_label")
    ref = []
    for im in batches:
        o = pipe(im)
        torch.cuda.synchronize()
        ref.append({k: o[k].clone() for k in keys})
    assert len({tuple(r["crop_list"].flatten().tolist()) for r in ref}) > 1        # the batches really differ
    keep = [{k: torch.empty_like(v) for k, v in r.items()} for r in ref]
    torch.cuda.synchronize()
    runner = PipelinedRunner(pipe)
    for rnd in range(3):
        for i, im in enumerate(batches):
            o = runner.submit(im)
            with torch.cuda.stream(runner.s_cls):
                for k in keys:
                    keep[i][k].copy_(o[k])
            del o                                                   # like `out = step(images)` in a loop
        runner.sync()
        for i, r in enumerate(ref):
            for k in keys:
                assert torch.equal(keep[i][k], r[k]), (rnd, i, k)


def test_pipelined_runner_with_fresh_inputs_dropped_after_submit():
    """A streaming caller: every batch is a NEW device tensor that the caller drops right after submit().  The runner
    reads it on the detect / classify / sub streams later, so it must be record_stream-ed there - otherwise the caching
    allocator hands its block to the next batch's upload while the previous batch's kernels are still queued."""
    from yvhip import engines
    from yvhip.pipeline import DetectClassifyPipeline, PipelinedRunner
    name, S, B = "vit_tiny_test", 128, 4
    pipe = DetectClassifyPipeline(engines.YoloEngine(engines.init_yolo_state("n", 5, 3, 4.0), "n", 5, S, DEV),
                                  [engines.VitEngine(engines.init_vit_wrapper_state(name, 5, 4), name, 5, device=DEV)],
                                  max_crops_per_image=3)
    g = torch.Generator().manual_seed(41)
    host = [torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8).pin_memory() for _ in range(8)]
    keys = ("crop_list", "crop_total", "cls_logits", "cls_label")
    ref = []
    for im in host:
        o = pipe(im.to(DEV))
        torch.cuda.synchronize()
        ref.append({k: o[k].clone() for k in keys})
    keep = [{k: torch.empty_like(v) for k, v in r.items()} for r in ref]
    torch.cuda.synchronize()
    for split in (False, True):
        runner = PipelinedRunner(pipe, split_classifier=split)
        for rnd in range(3):
            for i, im in enumerate(host):
                o = runner.submit(im.to(DEV, non_blocking=True))      # temporary: freed as soon as submit returns
                with torch.cuda.stream(runner.s_cls):
                    for k in keys:
                        keep[i][k].copy_(o[k])
                del o
            runner.sync()
            for i, r in enumerate(ref):
                for k in keys:
                    assert torch.equal(keep[i][k], r[k]), (split, rnd, i, k)


def test_step_is_graph_capturable():
    """The whole detect -> NMS -> crop -> classify step has no host synchronisation (the crop count stays on the
    device), so it can be captured into one hipGraph and replayed on new images: results equal the eager step."""
    from yvhip import engines
    from yvhip.pipeline import DetectClassifyPipeline
    name, S, B = "vit_tiny_test", 128, 4
    pipe = DetectClassifyPipeline(engines.YoloEngine(engines.init_yolo_state("n", 5, 3, 4.0), "n", 5, S, DEV),
                                  [engines.VitEngine(engines.init_vit_wrapper_state(name, 5, 4), name, 5, device=DEV)],
                                  max_crops_per_image=3)
    g = torch.Generator().manual_seed(31)
    batches = [torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8).to(DEV) for _ in range(4)]
    keys = ("det_count", "det_box", "crop_list", "crop_total", "cls_logits", "cls_label")
    ref = []
    for im in batches:
        o = pipe(im)
        torch.cuda.synchronize()
        ref.append({k: o[k].clone() for k in keys})
    static = batches[0].clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                                  # warm the capture stream's workspace registration
        pipe(static)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = pipe(static)
    for im, r in zip(batches, ref):
        static.copy_(im)
        graph.replay()
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(out[k], r[k]), k


def test_pipeline_with_zero_detections():
    """No candidate above the score threshold: every later stage sees an empty batch (device-side count 0)."""
    from yvhip import engines
    from yvhip.pipeline import DetectClassifyPipeline
    name, S, B = "vit_tiny_test", 64, 2
    ysd = engines.init_yolo_state("n", 5, 3, 1.0, cls_bias=-30.0)
    pipe = DetectClassifyPipeline(engines.YoloEngine(ysd, "n", 5, S, DEV),
                                  [engines.VitEngine(engines.init_vit_wrapper_state(name, 5, 4), name, 5, device=DEV)])
    img = torch.zeros(B, S, S, 3, dtype=torch.uint8, device=DEV)
    out = pipe(img)
    torch.cuda.synchronize()
    assert out["num_dets"].cpu().tolist() == [[0], [0]] and int(out["crop_total"][0]) == 0
    assert out["det_count"].cpu().tolist() == [0, 0]
    assert torch.all(out["cls_label"] == -1) and float(out["cls_logits"].abs().sum()) == 0


def test_vit_large_engine_shapes():
    """ViT-L/16 (BASELINE configs[4] classifier): D = 1024, 16 heads, 24 blocks through the same kernels."""
    from yvhip import engines
    sd = engines.init_vit_wrapper_state("vit_large_patch16_224", 5, seed=1)
    assert sum(v.numel() for k, v in sd.items() if k.startswith("model.")) == 304326632
    eng = engines.VitEngine(sd, "vit_large_patch16_224", 5)
    g = torch.Generator().manual_seed(0)
    pm = (torch.rand(2 * 196, 768, generator=g) * 2 - 1).to(torch.bfloat16).to(DEV)
    feats = eng.backbone(pm, 2)
    torch.cuda.synchronize()
    assert feats.shape == (2, 1024) and bool(torch.isfinite(feats).all()) and float(feats[:, :1000].abs().sum()) > 0


@pytest.mark.parametrize("c,n,B,H,W", [(16, 1, 2, 40, 24), (32, 2, 2, 32, 32), (32, 1, 1, 17, 33), (16, 2, 1, 16, 16),
                                       (16, 1, 2, 160, 160), (32, 2, 3, 80, 80), (32, 2, 1, 5, 70)])
def test_c2f_fused_equals_layer_by_layer(c, n, B, H, W):
    """yv_c2f_fused (one launch, intermediates in LDS: tile + halo, zero outside the image) against the same block run as the
    engine runs it layer by layer (yv_conv2d x (2 + 2n): cv1, the bottlenecks' 3 x 3 pairs with the bf16 residual, cv2).  Same K
    order, f32 accumulation and rounding chain, so the two must agree to the last bit or, where the layer kernels split K
    differently, to one bf16 ulp of a few outputs; shapes cover ragged tiles (H, W not multiples of 16), images smaller than a
    tile and than the halo, both channel widths and depths."""
    import yvhip as yv
    g = torch.Generator().manual_seed(c * 100 + n * 10 + H)
    C1 = 2 * c
    mk = lambda co, k: ((torch.randn(co, k, generator=g) * (2.0 / k) ** 0.5).to(torch.bfloat16).to(DEV),
                        (torch.randn(co, generator=g) * 0.1).to(DEV))
    w1, b1 = mk(C1, C1)
    wm = [mk(c, 9 * c) for _ in range(2 * n)]
    w2, b2 = mk(C1, (2 + n) * c)
    x = torch.randn(B, H, W, C1, generator=g).to(torch.bfloat16).to(DEV)
    # layer by layer
    y = torch.zeros(B, H, W, (2 + n) * c, dtype=torch.bfloat16, device=DEV)
    t = torch.zeros(B, H, W, c, dtype=torch.bfloat16, device=DEV)
    ref = torch.zeros(B, H, W, C1, dtype=torch.bfloat16, device=DEV)
    yv.conv2d(yv.view(x, 0, C1), None, B, H, W, 1, 1, w1, b1, y, 0, yv.EPI_SILU)
    for j in range(n):
        src = (1 + j) * c
        yv.conv2d(yv.view(y, src, c), None, B, H, W, 3, 1, wm[2 * j][0], wm[2 * j][1], t, 0, yv.EPI_SILU)
        yv.conv2d(yv.view(t, 0, c), None, B, H, W, 3, 1, wm[2 * j + 1][0], wm[2 * j + 1][1], y, src + c,
                  yv.EPI_SILU | yv.EPI_RES_BF16, res=y, res_c_off=src)
    yv.conv2d(yv.view(y, 0, (2 + n) * c), None, B, H, W, 1, 1, w2, b2, ref, 0, yv.EPI_SILU)
    out = torch.full((B, H, W, C1), 7.0, dtype=torch.bfloat16, device=DEV)
    yv.c2f_fused(x, c, n, w1, b1, [m[0] for m in wm], [m[1] for m in wm], w2, b2, out)
    torch.cuda.synchronize()
    a, r = out.float().cpu(), ref.float().cpu()
    assert torch.isfinite(a).all()
    differ = int((a != r).sum())
    print(f"c2f c={c} n={n} {B}x{H}x{W}: {differ} of {a.numel()} outputs differ, rel-L2 {rel_l2(a, r):.2e}")
    assert rel_l2(a, r) < 2e-3
    assert torch.allclose(a, r, atol=2e-2, rtol=2e-2)
    assert differ <= a.numel() // 50


@pytest.mark.parametrize("scale", ["n", "s"])
def test_yolo_engine_fused_c2f_vs_layer_by_layer(scale):
    """The detector with its backbone C2f blocks fused (model.2, model.4 of YOLOv8n; model.2 of YOLOv8s: c = 32, n = 1) against
    the same engine running them layer by layer: raw head outputs within the bf16 noise of a few re-rounded intermediates."""
    from yvhip import engines
    sd = oy.init_state(scale, 5, seed=7)
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (2, 320, 320, 3), generator=g, dtype=torch.uint8).to(DEV)
    eng = engines.YoloEngine(sd, scale, 5, 320)
    assert eng.fused_c2f
    box_f, cls_f = eng.forward_raw(img)
    f2, f4 = eng._buffers(2)["out"][2].clone(), eng._buffers(2)["out"][4].clone()
    eng.fused_c2f = False
    box_u, cls_u = eng.forward_raw(img)
    u2, u4 = eng._buffers(2)["out"][2], eng._buffers(2)["out"][4]
    torch.cuda.synchronize()
    assert rel_l2(f2.float().cpu(), u2.float().cpu()) < 2e-3 and rel_l2(f4.float().cpu(), u4.float().cpu()) < 4e-3
    for a, b in zip(box_f + cls_f, box_u + cls_u):
        assert rel_l2(a.float().cpu(), b.float().cpu()) < 1e-2
