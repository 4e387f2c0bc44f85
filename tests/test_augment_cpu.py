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
