"""GPU parity of `yv_augment_patchify` (csrc/augment.hip) against oracle/augment.py: bit-exact bf16 rows for drawn
records of data_transforms['train'] (utils/trainClass.py:199-216), edge records and hostile tables."""
import numpy as np
import pytest
import torch

from oracle import augment as oa

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def yv():
    import yvhip
    yvhip.require_gpu()
    return yvhip


def _run(yv, x, geo, idx, P):
    out = yv.augment_patchify(torch.from_numpy(x).to(DEV), torch.from_numpy(geo).to(DEV), torch.from_numpy(idx).to(DEV), P)
    torch.cuda.synchronize()
    return out.float().cpu().numpy()


def _check(yv, x, geo, idx, P):
    got = _run(yv, x, geo, idx, P)
    g2 = (x.shape[2] // P) ** 2
    for b in range(x.shape[0]):
        want = oa.apply_record(x[b], geo[b], idx[b], P)
        np.testing.assert_array_equal(got[b * g2:(b + 1) * g2], want, err_msg=f"sample {b}")


@pytest.mark.parametrize("S,P", [(224, 16), (224, 8), (64, 32)])
def test_drawn_records_bit_exact(yv, S, P):
    from yvhip.augment import TrainAugment
    B = 48
    x = np.random.default_rng(S + P).standard_normal((B, 3, S, S)).astype(np.float32)
    geo, idx = TrainAugment(S, seed=11).sample(B)
    assert (idx[:, 3] > 0).any() and (np.abs(geo[:, 0] - 1) > 1e-3).any()          # the draw exercises holes and warps
    _check(yv, x, geo, idx, P)


def test_every_transform_at_once(yv):
    from yvhip.augment import make_record
    S, P = 224, 16
    rng = np.random.default_rng(5)
    x = rng.standard_normal((4, 3, S, S)).astype(np.float32)
    recs = [
        make_record(S, flip=True, crop_xy=(24, 24), ssr=(10.0, 1.05, 0.0625, -0.0625), perm=(2, 0, 1),
                    grid=(1 + rng.uniform(-.05, .05, 6), 1 + rng.uniform(-.05, .05, 6)),
                    holes=[(0, 0, 11, 11), (213, 213, 224, 224), (5, 100, 16, 111), (100, 5, 111, 16), (60, 60, 71, 71)]),
        make_record(S, flip=False, crop_xy=(0, 0), ssr=(-10.0, 0.95, -0.0625, 0.0625), elastic=rng.uniform(-50, 50, (3, 2)),
                    holes=[(i * 20, i * 25, i * 20 + 11, i * 25 + 11) for i in range(8)]),
        make_record(S, elastic=np.full((3, 2), 50.0)),
        make_record(S, ssr=(0.0, 1.0, 0.0, 0.0)),
    ]
    geo, idx = np.stack([r[0] for r in recs]), np.stack([r[1] for r in recs])
    _check(yv, x, geo, idx, P)


def test_identity_equals_patchify(yv):
    from yvhip.augment import identity_record
    from yvhip.modules import patchify_bf16
    S, P, B = 224, 16, 3
    x = torch.randn(B, 3, S, S, device=DEV)
    g, i = identity_record(S)
    geo = torch.from_numpy(np.stack([g] * B)).to(DEV)
    idx = torch.from_numpy(np.stack([i] * B)).to(DEV)
    assert torch.equal(yv.augment_patchify(x, geo, idx, P), patchify_bf16(x, P))


def test_hostile_records_stay_in_bounds(yv):
    """Tables and matrices come from the host: NaN / huge matrices, out-of-range table entries, channel ids and hole
    counts must neither fault nor read outside the sample (the oracle states the same clamps)."""
    from yvhip.augment import identity_record
    S, P = 64, 8
    x = np.random.default_rng(0).standard_normal((4, 3, S, S)).astype(np.float32)
    recs = [identity_record(S) for _ in range(4)]
    recs[0][0][0:6] = (np.nan, 1e30, -1e30, np.inf, 0.0, np.nan)
    recs[1][1][36:] = np.random.default_rng(1).integers(-10**6, 10**6, 2 * S)
    recs[2][1][0:4] = (7, -3, 2, 1000)
    recs[2][1][4:36] = np.random.default_rng(2).integers(-500, 500, 32)
    recs[3][0][6:] = np.random.default_rng(3).uniform(-1e6, 1e6, 2 * S)
    geo, idx = np.stack([r[0] for r in recs]), np.stack([r[1] for r in recs])
    _check(yv, x, geo, idx, P)


def test_argument_checks(yv):
    x = torch.zeros(1, 3, 64, 64, device=DEV)
    geo = torch.zeros(1, 6 + 128, device=DEV)
    idx = torch.zeros(1, 36 + 128, dtype=torch.int32, device=DEV)
    with pytest.raises(yv.YvError):
        yv.augment_patchify(x, geo[:, :-1].contiguous(), idx, 16)
    with pytest.raises(yv.YvError):
        yv.augment_patchify(x, geo, idx, 12)                      # patch must be a multiple of 8 dividing S
    with pytest.raises(yv.YvError):
        yv.augment_patchify(x.cpu(), geo, idx, 16)


def test_train_one_epoch_applies_the_device_stage(yv, monkeypatch, tmp_path):
    """train_one_epoch must route a loader built with data_transforms['train'] through yv_augment_patchify."""
    from utils import trainClass as tc
    from utils.class_config import CFG
    calls = []
    real = yv.augment_patchify

    def spy(x, geo, idx, patch, out=None):
        calls.append((tuple(x.shape), tuple(geo.shape)))
        return real(x, geo, idx, patch, out)

    monkeypatch.setattr(yv, "augment_patchify", spy)
    t = tc.build_transforms(CFG)["train"]

    class DS(torch.utils.data.Dataset):
        transforms = t

        def __len__(self):
            return 8

        def __getitem__(self, i):
            img = np.random.default_rng(i).integers(0, 256, (40 + i, 30 + i, 3), dtype=np.uint8)
            chw = torch.from_numpy(np.ascontiguousarray(np.transpose(t(image=img)["image"], (2, 0, 1))))
            return chw, torch.nn.functional.one_hot(torch.tensor(i % CFG.num_classes), CFG.num_classes), "p"

    loader = torch.utils.data.DataLoader(DS(), batch_size=4)
    tc.set_seed(0)
    from yvhip import engines
    wpath = str(tmp_path / "w.pth")
    torch.save(engines.init_vit_wrapper_state("vit_tiny_test", CFG.num_classes, seed=8), wpath)
    net = tc.build_model(CFG, pretrained=wpath, modelName="vit_tiny_test").to(DEV)
    correct = tc.train_one_epoch(net, None, loader, tc.build_loss, None, [0.01], 4, 0, 2, True, DEV)
    assert len(calls) == 2 and calls[0] == ((4, 3, 224, 224), (4, 6 + 448))
    assert 0 <= correct <= 8


# ------------------------------------------------------------------------------------------- detector augmentation
def _mosaic_inputs(S, B, n_tiles, seed):
    from yvhip.yolo_augment import DetAugment, build_record, tile_geometry
    rng = np.random.default_rng(seed)
    raw = [(int(rng.integers(S // 3, 2 * S)), int(rng.integers(S // 3, 2 * S))) for _ in range(n_tiles)]
    sizes = [tile_geometry(w, h, S) for w, h in raw]
    tiles = np.full((n_tiles, S, S, 3), 114, np.uint8)
    for k, (w, h) in enumerate(sizes):
        tiles[k, :h, :w] = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    aug = DetAugment(S, seed=seed)
    rf, ri, lut = [], [], []
    for b in range(B):
        p = aug.plan(b % n_tiles, n_tiles, use_mosaic=(b % 4 != 3))
        f, i, l, _, _, _ = build_record(p, [sizes[s] for s in p["sources"]], p["sources"], S)
        rf.append(f); ri.append(i); lut.append(l)
    return tiles, np.stack(rf), np.stack(ri), np.stack(lut)


@pytest.mark.parametrize("S,B", [(64, 16), (640, 2)])
def test_mosaic_kernel_bit_exact(yv, S, B):
    from oracle import yolo_augment as oy
    tiles, rf, ri, lut = _mosaic_inputs(S, B, 6, seed=S)
    out = yv.mosaic_augment(*(torch.from_numpy(a).to(DEV) for a in (tiles, rf, ri, lut)))
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    assert out.shape == (B, S, S, 3)
    for b in range(B):
        np.testing.assert_array_equal(out[b], oy.apply_record(tiles, rf[b], ri[b], lut[b], S), err_msg=f"image {b}")


def test_mosaic_kernel_hostile_records(yv):
    from oracle import yolo_augment as oy
    S, B = 32, 4
    tiles, rf, ri, lut = _mosaic_inputs(S, B, 4, seed=9)
    rf[0] = (np.nan, 1e30, -1e30, np.inf, 0, 1)
    ri[1, 2::8] = (99, -5, 4, 7)                                   # tile ids outside the tile array
    ri[2, 0] = 1000
    ri[2, 2:] = np.random.default_rng(0).integers(-200, 200, 32)
    ri[2, 2::8] = (0, 1, 2, 3)
    lut[3] = np.random.default_rng(1).integers(0, 256, (3, 256), dtype=np.uint8)
    out = yv.mosaic_augment(*(torch.from_numpy(a).to(DEV) for a in (tiles, rf, ri, lut))).cpu().numpy()
    for b in range(B):
        np.testing.assert_array_equal(out[b], oy.apply_record(tiles, rf[b], ri[b], lut[b], S), err_msg=f"image {b}")


def test_augment_batch_from_files(yv, tmp_path):
    """Files -> tiles (yv_letterbox) -> composed batch: labels stay inside the image and follow their objects: a bright
    rectangle on a dark image must still be bright inside its transformed box."""
    from PIL import Image
    from yvhip.yolo_augment import DetAugment, augment_batch
    S, n = 128, 5
    rng = np.random.default_rng(0)
    (tmp_path / "images").mkdir(); (tmp_path / "labels").mkdir()
    samples = []
    for i in range(n):
        w, h = int(rng.integers(90, 200)), int(rng.integers(90, 200))
        img = np.full((h, w, 3), 20, np.uint8)
        x0, y0, bw, bh = int(rng.integers(5, w // 2)), int(rng.integers(5, h // 2)), w // 3, h // 3
        img[y0:y0 + bh, x0:x0 + bw] = 235
        ip, lp = tmp_path / "images" / f"a{i}.png", tmp_path / "labels" / f"a{i}.txt"
        Image.fromarray(img).save(ip)
        lp.write_text(f"{i % 3} {(x0 + bw / 2) / w} {(y0 + bh / 2) / h} {bw / w} {bh / h}\n")
        samples.append((str(ip), str(lp)))
    aug = DetAugment(S, seed=4)
    checked = 0
    for use_mosaic in (True, False):
        img, gtb, gtl, gtn = augment_batch(samples, range(4), aug, 8, DEV, use_mosaic=use_mosaic)
        assert img.shape == (4, S, S, 3) and img.dtype == torch.uint8 and img.is_cuda
        im = img.cpu().numpy().astype(np.float32).max(axis=3)      # the value channel survives the hue / saturation gains
        for b in range(4):
            for j in range(int(gtn[b])):
                x1, y1, x2, y2 = gtb[b, j].tolist()
                assert 0 <= x1 < x2 <= S and 0 <= y1 < y2 <= S and 0 <= int(gtl[b, j]) < 3
                if x2 - x1 >= 6 and y2 - y1 >= 6:
                    inner = im[b, int(y1) + 2:int(y2) - 2, int(x1) + 2:int(x2) - 2]
                    assert inner.mean() > 100, (use_mosaic, b, j, inner.mean())
                    checked += 1
    assert checked >= 6
