"""GPU: the reference's call surface end to end (app.py / test.py usage pattern) on the HIP path."""
import os

import numpy as np
import pytest
import torch

from oracle import boxes as ob
from oracle import pipeline as op
from oracle import train as ot
from oracle import vit as ov

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _CFG:
    num_classes, device, modelName, pretrained = 5, DEV, "vit_tiny_test", None


def _make_images(tmp_path, sizes):
    from PIL import Image
    g = np.random.default_rng(5)
    paths = []
    for i, (w, h) in enumerate(sizes):
        p = str(tmp_path / f"img_{i}.png")
        Image.fromarray(g.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(p)
        paths.append(p)
    return paths


def test_trtmodule_contract():
    from YOLOTensorRT.models import TRTModule
    eng = TRTModule("random:n:5:1", torch.device(DEV), size=128)
    eng.set_desired(['num_dets', 'bboxes', 'scores', 'labels'])
    H, W = eng.inp_info[0].shape[-2:]
    assert (H, W) == (128, 128)
    x = torch.rand(1, 3, 128, 128)
    x = (x * 255).round() / 255                                  # a blob: u8 / 255
    num, bb, sc, lb = eng(x.to(DEV))
    # KAT-2 (test.ipynb:20-24): shapes and dtypes of the 4 engine outputs
    assert num.shape == (1, 1) and num.dtype == torch.int32
    assert bb.shape == (1, 100, 4) and bb.dtype == torch.float32
    assert sc.shape == (1, 100) and sc.dtype == torch.float32 and lb.shape == (1, 100) and lb.dtype == torch.int32
    n = int(num[0, 0])
    assert torch.all(sc[0, :n] > 0.25) and float(sc[0, n:].abs().sum()) == 0


def test_main_end_to_end(tmp_path):
    import utils.utils as uu
    from YOLOTensorRT.inferdet import draw_image, main
    from YOLOTensorRT.models import TRTModule
    from yvhip import engines
    S = 128
    eng = TRTModule("random:n:5:1", torch.device(DEV), size=S)
    vsd = engines.init_vit_wrapper_state("vit_tiny_test", 5, seed=2)
    wpath = str(tmp_path / "best.pth")
    torch.save(vsd, wpath)
    model_list = [uu.build_model(CFG=_CFG, modelName="vit_tiny_test", pretrained=wpath)]
    model_list[-1].to(DEV); model_list[-1].eval()
    sizes = [(128, 128), (200, 150), (97, 131)]
    paths = _make_images(tmp_path, sizes)
    seen = []
    res = main(Engine=eng, imgs=str(tmp_path), device=torch.device(DEV), model_list=model_list,
               transform={"valid_test": None}, aliyunoss=None,
               func=lambda folder, name, path, objs: seen.append((name, len(objs))))
    import json
    json.dumps(res)                                               # jsonify-able (app.py:62)
    assert [r["image"] for r in res["output"]] == sorted(os.path.basename(p) for p in paths)
    assert sorted(seen) == sorted((r["image"], len(r["objects"])) for r in res["output"])
    from PIL import Image
    from YOLOTensorRT.models.utils import letterbox_geometry
    import yvhip
    total = 0
    for r_, p in zip(res["output"], sorted(paths)):
        img = np.asarray(Image.open(p).convert("RGB"))
        h, w = img.shape[:2]
        ratio, dwdh, (nw, nh), (left, top) = letterbox_geometry(h, w, (S, S))
        # re-run the detector on the device letterbox of this image, then the ORACLE post chain
        src = torch.from_numpy(img).to(DEV)[None].contiguous()
        geom = torch.tensor([[w, h, nw, nh, left, top]], dtype=torch.int32, device=DEV)
        net_in = yvhip.letterbox(src, geom, S)
        num, bb, sc, lb = [t.cpu() for t in eng(net_in)]
        dets = op.post_stages(num[0, 0], bb[0], sc[0], lb[0], float(np.float32(ratio)), dwdh, (w, h))
        dets = [d for d in dets if d["ok"]]
        assert len(r_["objects"]) == len(dets)
        for o, d in zip(r_["objects"], dets):
            assert [o["xmin"], o["ymin"], o["xmax"], o["ymax"]] == d["box"]
            x = torch.from_numpy(ob.crop_resize_normalize(img, d["rect"]))[None]
            ref = ov.wrapper_forward(vsd, x, "vit_tiny_test")[0]
            top2 = ref.topk(2).values
            if float(top2[0] - top2[1]) > 0.05 * float(ref.abs().max()):
                from YOLOTensorRT.config import CLASSES
                assert o["sort"] == CLASSES[int(ref.argmax())]
            total += 1
    assert total > 0
    out = draw_image(image=np.zeros((64, 64, 3), np.uint8), box=[5, 5, 40, 40], cls=1)
    assert out.shape == (64, 64, 3) and out.any()


def test_identity_letterbox_is_exact():
    import yvhip
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (1, 128, 128, 3), generator=g, dtype=torch.uint8).to(DEV)
    geom = torch.tensor([[128, 128, 128, 128, 0, 0]], dtype=torch.int32, device=DEV)
    assert torch.equal(yvhip.letterbox(img, geom, 128), img)
    # 2x downscale of a constant image stays constant; padding is 114
    img = torch.full((1, 100, 200, 3), 77, dtype=torch.uint8, device=DEV)
    geom = torch.tensor([[200, 100, 128, 64, 0, 32]], dtype=torch.int32, device=DEV)
    out = yvhip.letterbox(img, geom, 128).cpu()
    assert torch.all(out[0, 32:96] == 77) and torch.all(out[0, :32] == 114) and torch.all(out[0, 96:] == 114)


def test_wrapper_module_forward_matches_oracle(tmp_path):
    import utils.trainClass as tc
    from yvhip import engines
    sd = engines.init_vit_wrapper_state("vit_tiny_test", 5, seed=4)
    p = str(tmp_path / "w.pth"); torch.save(sd, p)
    net = tc.build_model(_CFG, pretrained=p, modelName="vit_tiny_test").to(DEV).eval()
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(3, 3, 224, 224, generator=g) * 2 - 1).to(torch.bfloat16).float()
    with torch.no_grad():
        y = net(x.to(DEV))
    ref = ov.wrapper_forward(sd, x, "vit_tiny_test")
    assert y.shape == (3, 5) and float((y.cpu() - ref).norm() / ref.norm()) < 2e-2
    # weights change -> device copies are rebuilt
    with torch.no_grad():
        net.fc[3].bias.add_(1.0)
        y2 = net(x.to(DEV))
    assert torch.allclose(y2.cpu(), y.cpu() + 1.0, atol=1e-4)


def test_losses_autograd(golden):
    import utils.trainClass as tc
    for c in golden["G3_loss"]:
        x = torch.tensor(c["x"], device=DEV, requires_grad=True)
        y = torch.nn.functional.one_hot(torch.tensor(c["label"]), 5).float().to(DEV)
        assert abs(float(tc.LabelSmoothingCrossEntropy(0.1)(x, y)) - c["lsce"]) < 3e-6 * max(1, abs(c["lsce"]))
        assert abs(float(tc.FocalLoss()(x, y)) - c["focal"]) < 3e-6 * max(1, abs(c["focal"]))
        loss = tc.build_loss(x, y)
        (loss * 2.0).backward()
        assert torch.allclose(x.grad.cpu(), 2.0 * torch.tensor(c["grad"]), atol=1e-6, rtol=3e-5)


def test_train_one_epoch_surface(tmp_path):
    """utils.trainClass.train_one_epoch / valid_one_epoch with a plain loader (reference signature)."""
    import utils.trainClass as tc
    from yvhip import engines
    sd = engines.init_vit_wrapper_state("vit_tiny_test", 5, seed=8)
    p = str(tmp_path / "w.pth"); torch.save(sd, p)
    net = tc.build_model(_CFG, pretrained=p, modelName="vit_tiny_test").to(DEV)
    g = torch.Generator().manual_seed(3)
    xs = (torch.rand(6, 3, 224, 224, generator=g) * 2 - 1)
    ys = torch.nn.functional.one_hot(torch.tensor([0, 1, 2, 3, 4, 0]), 5)
    loader = [(xs[i:i + 2], ys[i:i + 2], ["p"] * 2) for i in (0, 2, 4)] + [(xs[:1], ys[:1], ["p"])]   # last one is short
    opt = torch.optim.SGD(net.parameters(), 0.01, momentum=0.9, weight_decay=1e-3)
    acc0, loss0 = tc.valid_one_epoch(net, tc.build_loss, loader[:3])
    before = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    for ep in range(3):
        tc.train_one_epoch(net, net, loader, tc.build_loss, opt, [0.01], 2, ep, 10, True, DEV)
    after = net.state_dict()
    assert any(not torch.equal(before[k], after[k].detach().cpu()) for k in before)
    acc1, loss1 = tc.valid_one_epoch(net, tc.build_loss, loader[:3])
    assert loss1 < loss0                                            # three epochs on 6 samples reduce the loss
    assert abs(opt.param_groups[0]["lr"] - tc.cosine_anneal_schedule(2, 10, 0.01)) < 1e-12


def test_train_yolo_surface(tmp_path):
    """utils.trainYolo.train(epochs, batch, data) (utils/trainYolo.py:6-35) on a dataset produced by the reference's own
    file formats: generate_annotation XML -> xml2pd -> images/labels tree -> data yaml -> 3 epochs of the device
    training step.  The loss must fall and the written weights must load back."""
    import random as _r
    from PIL import Image
    import numpy as np
    import utils.trainYolo as ty
    from utils.class_config import xml2pd
    from utils.utils import generate_annotation
    rng = np.random.default_rng(0)
    src = tmp_path / "new"; src.mkdir()
    for i in range(6):
        img = rng.integers(90, 130, (96, 128, 3), dtype=np.uint8)
        x0, y0 = int(rng.integers(8, 60)), int(rng.integers(8, 40))
        img[y0:y0 + 40, x0:x0 + 48] = (220, 40, 40) if i % 2 else (40, 40, 220)
        Image.fromarray(img).save(src / f"im{i}.png")
        generate_annotation("new", f"im{i}.png", f"im{i}.png",
                            [{"sort": i % 2, "xmin": x0, "ymin": y0, "xmax": x0 + 48, "ymax": y0 + 40}], save_dir=str(src) + "/")
    _r.seed(3)
    root = tmp_path / "yolo" / "fold0"
    xml2pd(str(src), yolo_root=str(root))
    n_train = len(list((root / "images" / "train").glob("*.png")))
    assert n_train >= 3
    (tmp_path / "config.yaml").write_text(f"path: {root}\ntrain: images/train\nval: images/val\nnc: 5\n"
                                          "names: ['good', 'broke', 'lose', 'uncovered', 'circle']\n")
    logs = []
    res = ty.train(epochs=3, batch=2, data=str(tmp_path / "config.yaml"), size=128, save=str(tmp_path / "w" / "best.pth"),
                   log=logs.append)
    assert len(res["epochs"]) == 3 and res["epochs"][0]["steps"] == n_train // 2 and isinstance(res["not_built"], list)
    n_val = len(list((root / "images" / "val").glob("*.png")))
    for v in (res["val_before"], res["val_after"]):
        assert v["images"] == n_val and v["instances"] == n_val and 0.0 <= v["map50_95"] <= v["map50"] <= 0.995
    assert res["optimizer"] == "adamw" and res["lr0"] == pytest.approx(0.001111) and res["accumulate"] == 32   # optimizer=auto
    assert all(np.isfinite(e["loss"]) for e in res["epochs"])
    assert res["epochs"][0]["lr"] == pytest.approx(0.001111) and res["epochs"][2]["lr"] < res["epochs"][1]["lr"]
    # validation after every epoch, best-fitness checkpoint next to the last one
    assert all("fitness" in e and 0.0 <= e["map50_95"] <= e["map50"] for e in res["epochs"])
    assert res["best_fitness"] == max(e["fitness"] for e in res["epochs"]) and 0 <= res["best_epoch"] < 3
    assert os.path.exists(str(tmp_path / "w" / "best_last.pth"))
    sd = torch.load(res["weights"], map_location="cpu", weights_only=True)
    assert "model.0.conv.weight" in sd and "model.22.cv3.2.2.bias" in sd and sd["model.0.bn.running_var"].shape == (16,)
    res2 = ty.train(epochs=1, batch=2, data=str(tmp_path / "config.yaml"), size=128, weights=res["weights"],
                    save=str(tmp_path / "w" / "again.pth"), log=logs.append, optimizer="SGD", lr0=1e-4)
    assert res2["epochs"][0]["steps"] == n_train // 2 and not any("random initialisation" in l for l in logs[-2:])
    assert res2["optimizer"] == "sgd_nesterov" and res2["lr0"] == 1e-4


def test_train_class_from_xml_directories(tmp_path):
    """utils.trainClass.train(CFG, log) (utils/trainClass.py:424-508) end to end on directories of VOC xml + images as
    app.py's annotation flow leaves them: deliver() 80/20 split -> xml2pd -> build_dataset / DataLoader -> epochs of
    the native fine-tune step -> best.pth + result.json."""
    import json as _json
    import random as _r
    import numpy as np
    from PIL import Image
    import utils.trainClass as tc
    from utils.utils import generate_annotation
    rng = np.random.default_rng(0)
    new = tmp_path / "new"; new.mkdir()
    for i in range(10):
        img = rng.integers(60, 200, (80, 100, 3), dtype=np.uint8)
        Image.fromarray(img).save(new / f"w{i}.png")
        objs = [{"sort": ["good", "broke", "lose", "uncovered", "circle"][i % 5], "xmin": 10, "ymin": 8, "xmax": 70, "ymax": 60}]
        if i % 3 == 0:
            objs.append({"sort": "circle", "xmin": 30, "ymin": 20, "xmax": 90, "ymax": 70})
        generate_annotation("new", f"w{i}.png", f"w{i}.png", objs, save_dir=str(new) + "/")
    (new / "orphan.png").write_bytes((new / "w0.png").read_bytes())                     # image without xml: skipped
    _r.seed(1)
    tc.deliver(str(new) + "/", str(tmp_path / "tr"), str(tmp_path / "va"))
    n_tr = len(list((tmp_path / "tr").glob("*.xml"))); n_va = len(list((tmp_path / "va").glob("*.xml")))
    assert n_tr + n_va == 10 and n_tr >= 5 and n_va >= 1 and (new / "orphan.png").exists()

    class C(_CFG):
        modelName = "vit_tiny_test"
        pretrained = str(tmp_path / "missing.pth")
        train_path = [str(tmp_path / "tr"), str(tmp_path / "does_not_exist")]
        valid_path = [str(tmp_path / "va")]
        epoch, lr, train_bs, valid_bs = 2, 5e-3, 2, 4
        img_size = [224, 224]
    objs, circ = tc.xml2pd(C.train_path)
    assert len(objs) + len(circ) == sum(1 + (int(p.stem[1:]) % 3 == 0) for p in (tmp_path / "tr").glob("*.png"))
    assert all(o["objects"]["label"] == 4 for o in circ) and all(o["objects"]["label"] != 4 for o in objs)
    ds = tc.build_dataset(objs, circ, val=True, transforms=tc.build_transforms(C)["valid_test"])
    x, y, path = ds[0]
    assert tuple(x.shape) == (3, 224, 224) and x.dtype == torch.float32 and int(y.sum()) == 1 and os.path.exists(path)
    res = tc.train(C, log=str(tmp_path / "result.json"), save_path=str(tmp_path / "out" / "best.pth"))
    assert sorted(res) == [1, 2] and all(0.0 <= r["val_acc"] <= 100.0 and np.isfinite(r["loss"]) for r in res.values())
    assert sorted(_json.load(open(tmp_path / "result.json"))) == ["1", "2"]
    if any(r["val_acc"] > 0 for r in res.values()):
        sd = torch.load(tmp_path / "out" / "best.pth", map_location="cpu", weights_only=True)
        assert "model.cls_token" in sd and "fc.3.weight" in sd


def test_main_from_concurrent_threads(tmp_path):
    """app.py:50-61,99-100 calls `main()` and the classifier modules from request threads without a lock.  Two threads
    that hammer `main()` over different image folders with the SAME engine objects (and a third that calls the wrapper
    module directly) must each get exactly what a single-threaded call returns: engine steps are serialised at the Python
    boundary (yvhip.guard.StepGuard)."""
    import threading
    import utils.utils as uu
    from YOLOTensorRT.inferdet import main
    from YOLOTensorRT.models import TRTModule
    from yvhip import engines
    S = 128
    eng = TRTModule("random:n:5:1", torch.device(DEV), size=S)
    vsd = engines.init_vit_wrapper_state("vit_tiny_test", 5, seed=2)
    wpath = str(tmp_path / "best.pth")
    torch.save(vsd, wpath)
    net = uu.build_model(CFG=_CFG, modelName="vit_tiny_test", pretrained=wpath)
    net.to(DEV); net.eval()
    folders = []
    for t in range(2):
        d = tmp_path / f"set{t}"
        d.mkdir()
        _make_images(d, [(128 + 8 * t, 128), (200, 150 + 16 * t), (97 + t, 131), (160, 160)])
        folders.append(str(d))
    g = torch.Generator().manual_seed(3)
    xs = torch.rand(3, 3, 224, 224, generator=g) * 2 - 1
    call = lambda f: main(Engine=eng, imgs=f, device=torch.device(DEV), model_list=[net], transform=None, aliyunoss=None)
    ref = [call(f) for f in folders]
    ref_logits = net(xs.to(DEV)).cpu()
    assert sum(len(r["objects"]) for rr in ref for r in rr["output"]) > 0
    errors = []

    def run_main(i):
        try:
            for _ in range(6):
                got = call(folders[i])
                if got != ref[i]:
                    errors.append(("main", i))
        except Exception as e:                       # noqa: BLE001 - surfaced below
            errors.append(("main-exc", i, repr(e)))

    def run_module():
        try:
            for _ in range(30):
                if not torch.equal(net(xs.to(DEV)).cpu(), ref_logits):
                    errors.append(("module",))
        except Exception as e:                       # noqa: BLE001
            errors.append(("module-exc", repr(e)))

    ts = [threading.Thread(target=run_main, args=(0,)), threading.Thread(target=run_main, args=(1,)),
          threading.Thread(target=run_module)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errors, errors[:4]
    assert eng.engine.guard.entries >= 2 * 6 + 2


@pytest.mark.parametrize("w,h,S", [(200, 150, 128), (150, 200, 128), (1920, 1080, 640), (333, 517, 640), (97, 131, 128),
                                   (640, 480, 640), (64, 48, 640)])
def test_letterbox_resample_vs_oracle(w, h, S):
    """Row A1 / N1: the device letterbox (bilinear, half-pixel centres, round half up, border 114) against
    oracle/boxes.py::letterbox_image on real non-identity geometries - down-scales, up-scales, both orientations.
    Bit-exact (same f32 operations, no contraction).  The RULE is this build's statement: cv2.resize's fixed-point
    INTER_LINEAR is not reproducible without OpenCV, so against the reference the resample stays parity-unpinned."""
    import yvhip
    from YOLOTensorRT.models.utils import letterbox_geometry
    g = np.random.default_rng(w * 7 + h)
    img = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
    # smooth + noisy content: gradients exercise the blend, noise the rounding
    yy, xx = np.mgrid[0:h, 0:w]
    img[..., 0] = ((xx * 255) // max(w - 1, 1)).astype(np.uint8)
    img[..., 1] = ((yy * 255) // max(h - 1, 1)).astype(np.uint8)
    ref = ob.letterbox_image(img, S, S)
    r, dwdh, (nw, nh), (left, top) = letterbox_geometry(h, w, (S, S))
    r2, dwdh2, (nw2, nh2), (top2, _, left2, _) = ob.letterbox_params(h, w, S, S)
    assert (nw, nh, left, top) == (nw2, nh2, left2, top2) and r == r2 and tuple(dwdh) == tuple(dwdh2)
    # two images of different size share one canvas (the batch path of inferdet.main)
    canvas = np.zeros((2, h + 5, w + 9, 3), dtype=np.uint8)
    canvas[0, :h, :w] = img
    canvas[1, :h, :w] = img[::-1, ::-1]
    geom = torch.tensor([[w, h, nw, nh, left, top]] * 2, dtype=torch.int32, device=DEV)
    out = yvhip.letterbox(torch.from_numpy(canvas).to(DEV), geom, S).cpu().numpy()
    assert np.array_equal(out[0], ref)
    assert np.array_equal(out[1], ob.letterbox_image(np.ascontiguousarray(img[::-1, ::-1]), S, S))
