"""CPU tests of the training-augmentation records (yvhip/augment.py) against the oracle's stage-by-stage statement
(oracle/augment.py): conventions of the composed inverse map, parameter ranges and probabilities of
data_transforms['train'] (utils/trainClass.py:199-216).  No GPU, no compute calls into the library."""
import numpy as np
import pytest

from oracle import augment as oa


def _img(S, seed=0):
    return np.random.default_rng(seed).standard_normal((3, S, S)).astype(np.float32)


def _unpatch(rows, S, P):
    g = S // P
    return rows.reshape(g, g, 3, P, P).transpose(2, 0, 3, 1, 4).reshape(3, S, S)


def _bf(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).to(torch.float32).numpy()


def test_identity_record_is_patchify():
    from yvhip.augment import identity_record
    S, P = 64, 16
    x = _img(S)
    geo, idx = identity_record(S)
    np.testing.assert_array_equal(_unpatch(oa.apply_record(x, geo, idx, P), S, P), _bf(x))


@pytest.mark.parametrize("flip,crop_xy,shift,perm", [
    (True, None, (0, 0), (0, 1, 2)),
    (False, (3, 17), (0, 0), (2, 0, 1)),
    (True, (24, 0), (0, 0), (1, 0, 2)),
    (False, None, (5, -9), (0, 1, 2)),
    (True, (11, 24), (-14, 3), (2, 1, 0)),
])
def test_record_equals_stage_by_stage(flip, crop_xy, shift, perm):
    """Integer-only parameter sets: the folded record must reproduce flip -> crop + reflect pad -> shift -> shuffle ->
    holes done one array operation at a time."""
    from yvhip.augment import make_record
    S, P = 224, 16
    x = _img(S, 1)
    holes = [(0, 0, 11, 11), (100, 213, 111, 224), (50, 60, 61, 71)]
    ssr = (0.0, 1.0, shift[0] / S, shift[1] / S) if shift != (0, 0) else None
    geo, idx = make_record(S, flip=flip, crop_xy=crop_xy, ssr=ssr, perm=perm, holes=holes)
    want = oa.sequential_integer(x, flip=flip, crop_xy=crop_xy, shift=shift, perm=perm, holes=holes)
    np.testing.assert_array_equal(_unpatch(oa.apply_record(x, geo, idx, P), S, P), _bf(want))


def test_rotation_direction_and_centre():
    """getRotationMatrix2D: a positive angle turns the picture counter-clockwise about (S/2 - .5, S/2 - .5)."""
    from yvhip.augment import make_record
    S, P = 32, 8
    x = _img(S, 2)
    geo, idx = make_record(S, ssr=(90.0, 1.0, 0.0, 0.0))
    got = _unpatch(oa.apply_record(x, geo, idx, P), S, P)
    np.testing.assert_array_equal(got, _bf(np.rot90(x, 1, axes=(1, 2))))


def test_scale_enlarges_about_the_centre():
    from yvhip.augment import ssr_matrix
    S = 224
    M = ssr_matrix(S, 0.0, 1.05, 0.0, 0.0)
    c = S / 2 - 0.5
    np.testing.assert_allclose(M @ [c, c, 1], [c, c, 1], atol=1e-9)
    np.testing.assert_allclose(M @ [c + 10, c, 1], [c + 10.5, c, 1], atol=1e-9)


def test_elastic_matrix():
    from yvhip.augment import elastic_matrix
    S = 224
    np.testing.assert_allclose(elastic_matrix(S, np.zeros((3, 2))), np.eye(3), atol=1e-12)
    j = np.array([[5.0, -3.0], [20.0, 7.0], [-50.0, 50.0]])
    M = elastic_matrix(S, j)
    c, q = S // 2, S // 3
    p1 = np.array([[c + q, c + q], [c + q, c - q], [c - q, c - q]], dtype=float)
    for a, b in zip(p1, p1 + j):
        np.testing.assert_allclose(M @ [a[0], a[1], 1], [b[0], b[1], 1], atol=1e-9)


def test_grid_lut_shape():
    """Cells of S//5 pixels; cell i spans steps[i] * cell source pixels; the partial last cell ends at S."""
    from yvhip.augment import grid_lut
    S = 224
    steps = np.array([1.05, 0.95, 1.0, 1.02, 0.97, 1.0])
    t = grid_lut(S, steps)
    assert t[0] == 0 and t[-1] == S
    assert np.all(np.diff(t) >= 0)
    cell, ends = 44, np.cumsum(steps[:5]) * 44
    for i in range(5):
        assert t[(i + 1) * cell - 1] == pytest.approx(ends[i])
        assert t[i * cell] == pytest.approx(ends[i - 1] if i else 0.0)


def test_draw_probabilities_and_ranges():
    from yvhip.augment import TrainAugment
    S = 224
    aug = TrainAugment(S, seed=7)
    n = 8000
    cnt = dict(flip=0, crop=0, ssr=0, perm=0, grid=0, elastic=0, holes=0)
    for _ in range(n):
        p = aug.draw()
        cnt["flip"] += p["flip"]
        if p["crop_xy"] is not None:
            cnt["crop"] += 1
            assert 0 <= p["crop_xy"][0] <= S - 200 and 0 <= p["crop_xy"][1] <= S - 200
        if p["ssr"] is not None:
            cnt["ssr"] += 1
            a, s, dx, dy = p["ssr"]
            assert abs(a) <= 10 and 0.95 <= s <= 1.05 and abs(dx) <= 0.0625 and abs(dy) <= 0.0625
        cnt["perm"] += p["perm"] != (0, 1, 2)
        assert sorted(p["perm"]) == [0, 1, 2]
        if p["grid"] is not None:
            cnt["grid"] += 1
            assert all(len(g) == 6 and np.all(np.abs(g - 1) <= 0.05) for g in p["grid"])
        if p["elastic"] is not None:
            cnt["elastic"] += 1
            assert p["grid"] is None and np.all(np.abs(p["elastic"]) <= 50)
        if p["holes"]:
            cnt["holes"] += 1
            assert 5 <= len(p["holes"]) <= 8
            for x1, y1, x2, y2 in p["holes"]:
                assert x2 - x1 == 11 and y2 - y1 == 11 and 0 <= x1 and x2 <= S and 0 <= y1 and y2 <= S
    want = dict(flip=.5, crop=.25, ssr=.25, perm=.5 * 5 / 6, grid=.125, elastic=.125, holes=.5)
    for k, w in want.items():
        assert abs(cnt[k] / n - w) < 0.02, (k, cnt[k] / n, w)


def test_sample_is_repeatable_and_well_formed():
    from yvhip.augment import TrainAugment
    S = 224
    a, b = TrainAugment(S, seed=3).sample(16), TrainAugment(S, seed=3).sample(16)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    geo, idx = a
    assert geo.shape == (16, 6 + 2 * S) and geo.dtype == np.float32
    assert idx.shape == (16, 36 + 2 * S) and idx.dtype == np.int32
    assert idx[:, 36:].min() >= 0 and idx[:, 36:].max() < S
    assert np.isfinite(geo).all()


def test_train_transform_carries_the_device_stage():
    from utils.class_config import CFG
    from utils.trainClass import build_transforms
    t = build_transforms(CFG)
    assert hasattr(t["train"], "device_augment") and not hasattr(t["valid_test"], "device_augment")
    img = np.random.default_rng(0).integers(0, 256, (37, 51, 3), dtype=np.uint8)
    np.testing.assert_array_equal(t["train"](image=img)["image"], t["valid_test"](image=img)["image"])


# ------------------------------------------------------------------------------------- detector augmentation (host side)
def _gray_tiles(sizes, S, seed=0):
    """Tiles the way yv_letterbox leaves them: the resized image in the top-left corner of an S x S slot, 114 elsewhere.
    Grey levels only: 8-bit HSV is the identity on them, so geometry can be compared exactly."""
    rng = np.random.default_rng(seed)
    tiles = np.full((len(sizes), S, S, 3), 114, dtype=np.uint8)
    ims = []
    for k, (w, h) in enumerate(sizes):
        g = rng.integers(0, 256, (h, w, 1), dtype=np.uint8).repeat(3, axis=2)
        tiles[k, :h, :w] = g
        ims.append(g)
    return tiles, ims


@pytest.mark.parametrize("centre,flip", [((64, 64), False), ((40, 90), True), ((95, 33), False), ((32, 32), True), ((96, 96), False)])
def test_mosaic_record_equals_explicit_canvas(centre, flip):
    """scale 1, translate (.5,.5): the output is the central S x S window of the published 2S canvas."""
    from oracle import yolo_augment as oy
    from yvhip.yolo_augment import build_record, hsv_tables
    S = 64
    sizes = [(64, 48), (40, 64), (64, 64), (30, 20)]
    tiles, ims = _gray_tiles(sizes, S)
    plan = dict(mosaic=True, sources=[0, 1, 2, 3], centre=centre, scale=1.0, translate=(0.5, 0.5), hsv=[1.0, 1.0, 1.0], flip=flip)
    rec_f, rec_i, lut, M, offs, canvas = build_record(plan, sizes, [0, 1, 2, 3], S)
    want, pads = oy.explicit_mosaic(ims, centre, S)
    want = want[S // 2:S // 2 + S, S // 2:S // 2 + S]
    if flip:
        want = want[:, ::-1]
    np.testing.assert_array_equal(oy.apply_record(tiles, rec_f, rec_i, lut, S), want)
    assert offs == pads and canvas == 2 * S
    np.testing.assert_array_equal(lut, hsv_tables([1, 1, 1]))


def test_single_image_record_is_centred_letterbox():
    from oracle import yolo_augment as oy
    from yvhip.yolo_augment import build_record
    S = 64
    tiles, ims = _gray_tiles([(64, 40)], S, 3)
    plan = dict(mosaic=False, sources=[0], scale=1.0, translate=(0.5, 0.5), hsv=[1.0, 1.0, 1.0], flip=False)
    rec_f, rec_i, lut, M, offs, canvas = build_record(plan, [(64, 40)], [0], S)
    want = np.full((S, S, 3), 114, np.uint8)
    want[12:52] = ims[0]
    np.testing.assert_array_equal(oy.apply_record(tiles, rec_f, rec_i, lut, S), want)
    assert offs == [(0, 12)] and canvas == S


def test_hsv_statement_against_colorsys():
    """The 8-bit HSV round trip with gains: within the 8-bit hue quantisation of the float conversion."""
    import colorsys
    from oracle import yolo_augment as oy
    from yvhip.yolo_augment import hsv_tables
    S = 16
    rng = np.random.default_rng(1)
    tiles = rng.integers(0, 256, (1, S, S, 3), dtype=np.uint8)
    tiles[0, 0, :4] = [(255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 255)]
    rec_i = np.zeros(34, np.int32)
    rec_i[0], rec_i[2:9] = 1, (0, 0, 0, S, S, 0, 0)
    ident = np.array([1, 0, 0, 0, 1, 0], np.float32)
    out = oy.apply_record(tiles, ident, rec_i, hsv_tables([1, 1, 1]), S)
    assert np.abs(out.astype(int) - tiles[0].astype(int)).max() <= 5          # hue is quantised to 2 degree steps: up to 255/60 levels
    np.testing.assert_array_equal(out[0, :4], tiles[0, 0, :4])
    gains = [1.0, 0.5, 0.8]
    out = oy.apply_record(tiles, ident, rec_i, hsv_tables(gains), S)
    for (y, x) in [(1, 1), (5, 9), (15, 15), (0, 0)]:
        r, g, b = (tiles[0, y, x] / 255.0)
        h, s, v = colorsys.rgb_to_hsv(r, g, b)
        want = np.array(colorsys.hsv_to_rgb(h, s * gains[1], v * gains[2])) * 255
        assert np.abs(out[y, x] - want).max() <= 6, (out[y, x], want)


def test_transform_boxes():
    from yvhip.yolo_augment import affine_matrix, transform_boxes
    S = 64
    M = affine_matrix(2 * S, S, 1.0, 0.5, 0.5)                          # window [32, 96) of the canvas
    boxes = np.array([[40, 40, 60, 70], [0, 0, 33, 33], [90, 90, 128, 128], [50, 50, 51.5, 80], [20, 60, 100, 64.5]], float)
    labels = np.arange(5)
    nb, nl = transform_boxes(boxes, labels, M, 1.0, S, flip=False)
    # box 1: 1 x 1 px left after clipping (too small); box 3: 1.5 px wide; box 2 keeps 6 x 6 of 38 x 38 (area < 10 %)
    assert nl.tolist() == [0, 4]
    np.testing.assert_allclose(nb[0], [8, 8, 28, 38])
    np.testing.assert_allclose(nb[1], [0, 28, 64, 32.5])
    fb, _ = transform_boxes(boxes, labels, M, 1.0, S, flip=True)
    np.testing.assert_allclose(fb[0], [36, 8, 56, 38])
    M2 = affine_matrix(2 * S, S, 0.5, 0.5, 0.5)                         # whole canvas at half size
    nb, _ = transform_boxes(boxes[:1], labels[:1], M2, 0.5, S, flip=False)
    np.testing.assert_allclose(nb[0], [20, 20, 30, 35])


def test_det_plan_ranges():
    from yvhip.yolo_augment import DetAugment, hsv_tables, tile_geometry
    S = 640
    aug = DetAugment(S, seed=5)
    flips = 0
    for k in range(2000):
        p = aug.plan(k % 7, 7)
        assert p["mosaic"] and len(p["sources"]) == 4 and p["sources"][0] == k % 7 and all(0 <= s < 7 for s in p["sources"])
        assert all(S // 2 <= c <= 3 * S // 2 for c in p["centre"])
        assert 0.5 <= p["scale"] <= 1.5 and all(0.4 <= t <= 0.6 for t in p["translate"])
        assert abs(p["hsv"][0] - 1) <= 0.015 and abs(p["hsv"][1] - 1) <= 0.7 and abs(p["hsv"][2] - 1) <= 0.4
        flips += p["flip"]
    assert abs(flips / 2000 - 0.5) < 0.04
    assert not aug.plan(0, 7, use_mosaic=False)["mosaic"]
    t = hsv_tables([1.01, 1.7, 0.6])
    assert t.shape == (3, 256) and t[0].max() < 180 and t[1][200] == 255 and t[2][100] == 60
    assert tile_geometry(1280, 720, 640) == (640, 360) and tile_geometry(100, 50, 640) == (640, 320)
    assert tile_geometry(640, 640, 640) == (640, 640) and tile_geometry(333, 500, 640) == (427, 640)
