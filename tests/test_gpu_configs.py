"""Full-size GPU parity of the BASELINE.json configurations (configs[1]..configs[4]) against the fp32 CPU oracle.

The kernel-level tests elsewhere use miniature models so that the oracle finishes in a blink; these run every
configuration's REAL model (ViT-B/16, ViT-L/16, YOLOv8n/s/m at 640 x 640) through the HIP path with a small batch -
the batch size does not change any kernel's code path except the tile count - and compare with the oracle on the same
seeded inputs.  Tolerances (north_star "stated fp tolerance"): bf16 path vs fp32 oracle rel-L2 <= 2e-2
on logits / raw head outputs, <= 3e-2 after the whole detect -> crop -> classify chain; integer stages bit-exact.
Parity is UNPINNED against timm / ultralytics (absent from the reference tree and from this image, see oracle/*.py).
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import boxes as ob
from oracle import pipeline as op
from oracle import vit as ov
from oracle import yolo as oy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def _patches(x, P):
    return torch.cat([torch.from_numpy(ob.patchify(x[r].numpy(), P)) for r in range(x.shape[0])]).to(torch.bfloat16)


# ------------------------------------------------------------------------------------- configs[1]: the headline
def test_config1_end_to_end_yolov8n_vitb16_640():
    """BASELINE.json configs[1] (YOLOv8n + ViT-B/16, 640 x 640, bf16) through the schedule bench.py times -
    PipelinedRunner with the detector overlap and the classifier split into two concurrent half-batches - at batch 8,
    three batches in flight.  Every stage is checked on the DEVICE's input to that stage:
      detector raw outputs / decoded boxes+scores   vs oracle/yolo.py           rel-L2 <= 2e-2, boxes <= 0.02*S px
      EfficientNMS of the device's decoded tensors   vs oracle/boxes.py          bit-exact
      restore / filter / int / custom_nms / inflate  vs oracle/pipeline.py       bit-exact (integers)
      crop list (batch assembly)                                                  bit-exact
      classifier logits of every crop                vs oracle/vit.py            rel-L2 <= 3e-2 per crop, labels where sure
    """
    from yvhip import engines
    from yvhip.pipeline import DetectClassifyPipeline, PipelinedRunner
    name, S, B, R = "vit_base_patch16_224", 640, 8, 4
    ysd = engines.init_yolo_state("n", 5, seed=42, head_gain=4.0)
    vsd = engines.init_vit_wrapper_state(name, 5, seed=42)
    yolo = engines.YoloEngine(ysd, "n", 5, S, DEV)
    pipe = DetectClassifyPipeline(yolo, [engines.VitEngine(vsd, name, 5, device=DEV)], max_crops_per_image=R)
    g = torch.Generator().manual_seed(1234)
    batches = [torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8) for _ in range(3)]
    dbatches = [b.to(DEV) for b in batches]
    runner = PipelinedRunner(pipe, split_classifier=True)
    outs = [runner.submit(b) for b in dbatches]
    runner.sync()
    keys = ("num_dets", "bboxes", "scores", "labels", "det_count", "det_box", "crop_rect", "crop_list", "crop_total",
            "cls_logits", "cls_label")
    outs = [{k: o[k].cpu() for k in keys} for o in outs]
    total_crops = 0
    for bi, (img, out) in enumerate(zip(batches, outs)):
        # detector: raw + decode vs the fp32 oracle (first batch only: 8 x 8.1 GFLOP on the CPU)
        gb, gs = yolo(dbatches[bi])
        torch.cuda.synchronize()
        if bi == 0:
            raw = oy.forward_raw(ysd, oy.blob(img), "n", 5)
            eb, es = oy.decode(raw, 5, S)
            assert rel_l2(gs.cpu(), es) < 2e-2
            assert float((gb.cpu() - eb).abs().max()) < 0.02 * S
        # NMS of the device's own decoded tensors: bit-exact
        num, bb, sc, lb = ob.efficient_nms(gb.cpu(), gs.cpu())
        assert torch.equal(out["num_dets"], num) and torch.equal(out["bboxes"], bb)
        assert torch.equal(out["scores"], sc) and torch.equal(out["labels"], lb)
        n_crops = 0
        for b in range(B):
            dets = op.post_stages(num[b, 0], bb[b], sc[b], lb[b], 1.0, (0.0, 0.0), (S, S), max_crops=R)
            assert int(out["det_count"][b]) == len(dets)
            for k, d in enumerate(dets):
                assert out["det_box"][b, k].tolist() == d["box"] and out["crop_rect"][b, k].tolist() == d["rect"]
                if not d["ok"]:
                    continue
                assert out["crop_list"][n_crops].tolist() == [b] + d["rect"] + [k]
                if bi == 0 or n_crops < 2:                    # classifier oracle: every crop of batch 0, two of the others
                    x = torch.from_numpy(ob.crop_resize_normalize(img[b].numpy(), d["rect"]))[None]
                    ref = ov.wrapper_forward(vsd, x, name)[0]
                    got = out["cls_logits"][n_crops]
                    assert rel_l2(got, ref) < 3e-2, (bi, b, k, got, ref)
                    top = ref.topk(2).values
                    if float(top[0] - top[1]) > 0.05 * float(ref.abs().max()):
                        assert int(out["cls_label"][n_crops]) == int(ref.argmax())
                n_crops += 1
        assert int(out["crop_total"][0]) == n_crops
        total_crops += n_crops
    assert total_crops >= 3 * B                      # the synthetic detector really produces work for the classifier


# ------------------------------------------------------------------------------------- configs[2]: ViT-B/16 fine-tune
def _oracle_grads(sd, x, labels, name):
    from oracle import train as ot
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits = ov.wrapper_forward(p, x, name)
    loss = ot.build_loss(logits, F.one_hot(labels.long(), 5).float())
    loss.backward()
    return loss.detach(), logits.detach(), {k: v.grad for k, v in p.items()}


def _relu_free_head(sd):
    """Network_Wrapper.fc = ReLU -> Linear(1000,128) -> ReLU -> Linear(128,nc) (utils/utils.py:64-70).  With random weights a
    handful of the 1000 + 128 pre-activations sit within bf16 noise of zero; whether the device's forward lands on the same
    side as the fp32 oracle is a coin flip per unit, and ONE flipped hidden unit changes every upstream gradient by ~1/sqrt(128):
    measured on identical kernels, the per-tensor gradient error against autograd then wanders between 3.5 % and 11 % with
    the rounding of the forward (tests/diagnostics/attn_forms.py).  Shifting both pre-activations well above zero makes the
    two ReLUs act as the identity on BOTH sides, so the comparison measures the kernels and not the coin flips; the masking
    itself is pinned by test_gpu_train.py::test_head_bwd (random masks, 1e-5 against autograd)."""
    sd = {k: v.clone() for k, v in sd.items()}
    sd["model.head.bias"] += 8.0                      # feats ~ N(0, 1.4^2) + 8 > 0
    sd["fc.1.weight"] *= 0.05
    sd["fc.1.bias"] += 6.0                            # hidden ~ 6 +- 0.6 > 0
    sd["fc.3.weight"] *= 0.1                          # logits stay O(1): the loss is not saturated
    return sd


def test_config2_vit_b16_trainer_vs_autograd():
    """BASELINE.json configs[2] model (ViT-B/16, 224 x 224) through VitTrainer forward + backward at R = 4 against fp32
    autograd of the oracle: logits rel-L2 <= 2e-2, loss within 2 %, and EVERY one of the 152 + 4 parameter gradients
    compared per tensor.  The error table is printed; the gate is the one the table supports (see GRAD_TOL)."""
    from yvhip.training import VitTrainer
    name, R = "vit_base_patch16_224", 4
    sd = _relu_free_head(ov.init_wrapper_state(name, seed=21))
    g = torch.Generator().manual_seed(R)
    x = (torch.rand(R, 3, 224, 224, generator=g) * 2 - 1).to(torch.bfloat16).float()
    labels = torch.randint(0, 5, (R,), generator=g, dtype=torch.int32)
    ref_loss, ref_logits, ref = _oracle_grads(sd, x, labels, name)
    tr = VitTrainer(sd, name, 5)
    pm = _patches(x, tr.P_).to(DEV)
    logits = tr.forward(pm, R)
    loss = tr.backward(pm, labels.to(DEV), R)
    torch.cuda.synchronize()
    assert rel_l2(logits.cpu(), ref_logits) < 2e-2
    assert abs(float(loss[0]) - float(ref_loss)) < 2e-2 * abs(float(ref_loss))
    got = tr.grad_dict()
    err = {k: rel_l2(got[k].cpu(), v) for k, v in ref.items()}
    worst = sorted(err.items(), key=lambda kv: -kv[1])
    vals = sorted(err.values())
    print("ViT-B/16 gradient rel-L2 vs fp32 autograd: median %.4f, max %.4f (%s)" % (vals[len(vals) // 2], worst[0][1], worst[0][0]))
    for k, e in worst[:6]:
        print("   %-44s %.4f" % (k, e))
    assert len(err) == 12 * 12 + 8 + 4
    assert worst[0][1] < GRAD_TOL, worst[:6]
    assert vals[len(vals) // 2] < GRAD_TOL_MEDIAN


# per-tensor gradient gates of the ViT trainer tests (rel-L2 against fp32 autograd); see test_config2_* for the table
# with the ReLU coin flips of the wrapper head taken out (_relu_free_head) what remains is the bf16 storage of every activation
# and activation gradient; measured on MI355X (round 2): median 0.0065, max 0.0123 (model.cls_token) over the 156 tensors
GRAD_TOL = 2.5e-2
GRAD_TOL_MEDIAN = 1.5e-2


# ------------------------------------------------------------------------------------- configs[4]: YOLOv8m + ViT-L/16
def test_config4_vit_large_vs_oracle_bf16_and_mxfp8():
    """ViT-L/16 (D 1024, 24 blocks, 16 heads) against oracle/vit.py at R = 2: the bf16 engine within rel-L2 2e-2 on the
    1000 backbone logits; the MXFP8 engine (block linears on e4m3 + E8M0, everything else bf16/f32) against the SAME fp32
    oracle within MX_TOL - stated for random-init weights, whose 1000 near-tied logits are the worst case for a 3-bit
    mantissa (see tests/test_gpu_fp8.py for the per-GEMM exactness statement)."""
    from yvhip import engines
    name, R = "vit_large_patch16_224", 2
    sd = ov.init_wrapper_state(name, seed=11)
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(R, 3, 224, 224, generator=g) * 2 - 1).to(torch.bfloat16).float()
    ref_feats = ov.vit_forward(sd, x, name)
    ref_logits = ov.wrapper_head(sd, ref_feats)
    pm = _patches(x, 16).to(DEV)
    cnt = torch.tensor([R], dtype=torch.int32, device=DEV)
    res = {}
    for dtype in ("bf16", "mxfp8"):
        eng = engines.VitEngine(sd, name, 5, dtype=dtype)
        buf = eng.patch_buffer(R)
        buf.copy_(pm)
        feats = eng.backbone(buf, R, cnt)
        logits = torch.zeros(R, 5, device=DEV); labels = torch.full((R,), -1, dtype=torch.int32, device=DEV)
        eng.head(feats, R, logits, labels, count=cnt)
        torch.cuda.synchronize()
        res[dtype] = (rel_l2(feats[:, :1000].cpu(), ref_feats), rel_l2(logits.cpu(), ref_logits),
                      float(F.cosine_similarity(feats[:, :1000].cpu().double(), ref_feats.double(), dim=1).min()))
        del eng
    print("ViT-L/16 vs fp32 oracle (feats rel-L2, wrapper logits rel-L2, min cosine): bf16 %s  mxfp8 %s" % (res["bf16"], res["mxfp8"]))
    assert res["bf16"][0] < 2e-2 and res["bf16"][1] < 2e-2
    assert res["mxfp8"][0] < MX_TOL and res["mxfp8"][2] > 0.99, res["mxfp8"]


MX_TOL = 0.15


def test_config4_yolov8m_640_vs_oracle():
    """YOLOv8m (nc 80 as published; the detector of configs[4]) at the full 640 x 640 resolution, batch 1."""
    from yvhip import engines
    scale, nc, size = "m", 80, 640
    sd = oy.init_state(scale, nc, seed=7)
    g = torch.Generator().manual_seed(2)
    img = torch.randint(0, 256, (1, size, size, 3), generator=g, dtype=torch.uint8)
    raw = oy.forward_raw(sd, oy.blob(img), scale, nc)
    eng = engines.YoloEngine(sd, scale, nc, size)
    box_l, cls_l = eng.forward_raw(img.to(DEV))
    torch.cuda.synchronize()
    a0 = 0
    for s, st in enumerate((8, 16, 32)):
        w = size // st
        part = raw[:, :, a0:a0 + w * w].reshape(1, 64 + nc, w, w)
        assert rel_l2(box_l[s].permute(0, 3, 1, 2).cpu(), part[:, :64]) < 2e-2, s
        assert rel_l2(cls_l[s][..., :nc].permute(0, 3, 1, 2).cpu(), part[:, 64:]) < 2e-2, s
        a0 += w * w
    eb, es = oy.decode(raw, nc, size)
    gb, gs = eng(img.to(DEV))
    assert rel_l2(gs.cpu(), es) < 2e-2
    assert float((gb.cpu() - eb).abs().max()) < 0.02 * size
