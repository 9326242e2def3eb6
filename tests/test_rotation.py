"""Rotation augmentation.  CPU: the product's geometry (neuroclear_amd/data/rotation.py) against
the oracle's restatement (oracle/rotate.py), reference-generated rectangles and closed-form cases; the oracle's restatement of
cv2.warpAffine's INTER_LINEAR arithmetic (OpenCV 4.5.0: 10 + 5 bit fixed-point coordinates, weights in 1/32 steps, integer blend for
uint8, float32 blend for uint16) against known answers that follow from the published algorithm by hand.  GPU: nc_rotate_crop through
SingleVolumeDataset against that restatement, BIT FOR BIT.  (OpenCV cannot be installed here: "restated from the published algorithm",
not pinned by cv2 outputs.)"""
import random
from argparse import Namespace

import numpy as np
import pytest

from neuroclear_amd.data import rotation
from oracle import rotate as orot


@pytest.mark.parametrize('h,w', [(40, 60), (108, 108), (33, 17)])
@pytest.mark.parametrize('angle', [0, 1, 30, 45, 90, 123, 180, 270, 359, -90, -270])
def test_geometry_matches_oracle(h, w, angle):
    inv, rect = rotation.rotate_clean_plan(h, w, angle)
    plan = orot.clean_rotation_plan(h, w, angle)
    np.testing.assert_allclose(inv, plan['inv'], rtol=0, atol=1e-9)
    assert rect == plan['rect']
    x1, y1, x2, y2 = rect
    assert 0 <= x1 < x2 <= plan['new_w'] and 0 <= y1 < y2 <= plan['new_h']


def test_geometry_matches_reference_fixture(golden_dir):
    """Inscribed-rectangle sizes and centre-crop rectangles produced by RUNNING the reference's own
    largest_rotated_rect / crop_around_center (oracle/gen_golden.py::gen_rotation): the product's closed form must land
    on the same integer rectangle for every whole degree, and within 1e-9 of the float size."""
    import math
    import os
    g = np.load(os.path.join(golden_dir, 'rotation_geometry.npz'))
    for w, h, a, rw, rh in g['rects']:
        w, h, a = int(w), int(h), int(a)
        got = rotation.inscribed_rect_size(w, h, math.radians(a))
        assert abs(got[0] - rw) < 1e-9 and abs(got[1] - rh) < 1e-9, (w, h, a)
        assert orot.largest_rotated_rect(w, h, math.radians(a)) == (rw, rh)  # the oracle restatement is bit-exact
        nw, nh, _ = rotation.canvas_and_affine(h, w, a)
        assert rotation.centered_rect(nw, nh, *got) == rotation.centered_rect(nw, nh, rw, rh), (w, h, a)
    for cw, ch, wd, ht, x1, y1, x2, y2 in g['crops']:
        want = (int(x1), int(y1), int(x2), int(y2))
        assert rotation.centered_rect(int(cw), int(ch), wd, ht) == want
        ox1, oy1, ox2, oy2 = orot.crop_rect(int(cw), int(ch), wd, ht)
        assert (max(ox1, 0), max(oy1, 0), min(ox2, int(cw)), min(oy2, int(ch))) == want


def test_geometry_closed_form():
    inv, rect = rotation.rotate_clean_plan(40, 60, 0)
    np.testing.assert_allclose(inv, [[1, 0, 0], [0, 1, 0]], atol=1e-12)
    assert rect == (0, 0, 60, 40)
    inv, rect = rotation.rotate_clean_plan(40, 60, 90)  # quarter turn: the canvas is the transposed slice, kept whole
    assert rect == (0, 0, 40, 60)
    v = (np.random.default_rng(0).random((2, 40, 60)) * 65535).astype(np.uint16)
    assert np.array_equal(orot.rotate_clean_3D_xy(v, 0), v)
    r = orot.rotate_clean_3D_xy(v, 90)
    assert r.shape == (2, 60, 40)
    # a quarter turn moves samples without interpolating: the multiset of values is (nearly) preserved
    assert abs(int(r.astype(np.int64).sum()) - int(v.astype(np.int64).sum())) <= 0.05 * v.astype(np.int64).sum()


def test_warp_affine_known_answers():
    """Cases whose cv2.warpAffine result follows from the published algorithm without running it.  With X = (cvRound(1024 (M' x
    terms)) + 16) >> 5 a source coordinate is rounded to the NEAREST 1/32 (ties up), and the weights are exactly (1 - f), f in 1/32 steps."""
    rng = np.random.default_rng(8)
    for dtype in (np.uint8, np.uint16):
        hi = np.iinfo(dtype).max
        src = rng.integers(0, hi + 1, (9, 14)).astype(dtype)
        eye = np.array([[1.0, 0, 0], [0, 1.0, 0]])
        assert np.array_equal(orot.warp_affine_cv(src, eye, 0, 0, 9, 14), src)
        # integer shift (+3, -2): dst(x, y) = src(x - 3, y + 2), zero outside
        sh = orot.warp_affine_cv(src, np.array([[1.0, 0, 3], [0, 1.0, -2]]), 0, 0, 9, 14)
        want = np.zeros_like(src)
        want[:7, 3:] = src[2:, :11]
        assert np.array_equal(sh, want)
        # half a pixel to the right: f = 16/32 on the pair (x - 1, x); uint16 rounds the float sum half-to-even, uint8 adds 2^14 and shifts
        half = orot.warp_affine_cv(src, np.array([[1.0, 0, 0.5], [0, 1.0, 0]]), 0, 0, 9, 14)
        a = np.concatenate([np.zeros((9, 1), np.int64), src[:, :-1].astype(np.int64)], axis=1)
        b = src.astype(np.int64)
        want = (a + b + 1) // 2 if dtype == np.uint8 else np.rint((a + b) / 2.0).astype(np.int64)
        assert np.array_equal(half.astype(np.int64), want)
        # 1/64 of a pixel: the source coordinate x - 1/64 is a tie between x - 1/32 and x and rounds up to x (no interpolation) ...
        assert np.array_equal(orot.warp_affine_cv(src, np.array([[1.0, 0, 1.0 / 64], [0, 1.0, 0]]), 0, 0, 9, 14), src)
        # ... while x + 1/64 rounds up to x + 1/32: weights 31/32 and 1/32 on (x, x + 1)
        q = orot.warp_affine_cv(src, np.array([[1.0, 0, -1.0 / 64], [0, 1.0, 0]]), 0, 0, 9, 14)
        nxt = np.concatenate([src[:, 1:].astype(np.int64), np.zeros((9, 1), np.int64)], axis=1)
        if dtype == np.uint8:
            want = (b * (32 * 32 * 31) + nxt * (32 * 32) + (1 << 14)) >> 15
        else:
            want = np.rint((b.astype(np.float32) * np.float32(31 / 32) + nxt.astype(np.float32) * np.float32(1 / 32)).astype(np.float64)).astype(np.int64)
        assert np.array_equal(q.astype(np.int64), want)
        # a quarter turn moves samples without interpolating.  The reference turns about (w/2, h/2) -- half a pixel off the centre of the
        # pixel grid -- so the turned image lands one row low: row 0 is border, the rest is np.rot90 without its last row
        sq = rng.integers(0, hi + 1, (12, 12)).astype(dtype)
        plan = orot.clean_rotation_plan(12, 12, 90)
        r = orot.warp_affine_cv(sq, plan['affine'], 0, 0, plan['new_h'], plan['new_w'])
        assert r.shape == (12, 12) and not r[0].any() and np.array_equal(r[1:], np.rot90(sq)[:-1])
    # the fixed-point result stays next to the textbook bilinear value: coordinates are off by at most 1/64 pixel per axis
    v = (rng.random((1, 40, 60)) * 65535).astype(np.uint16)
    plan = orot.clean_rotation_plan(40, 60, 33)
    x1, y1, x2, y2 = plan['rect']
    fixed = orot.warp_affine_cv(v[0], plan['affine'], x1, y1, y2 - y1, x2 - x1).astype(np.float64)
    exact = orot.warp_bilinear(v[0], plan['inv'], x1, y1, y2 - y1, x2 - x1).astype(np.float64)
    assert np.abs(fixed - exact).max() < 65535 * 2 / 64 and not np.array_equal(fixed, exact)


def test_product_inverse_is_warp_affines():
    for h, w, ang in ((40, 60, 33), (108, 108, 271), (33, 17, 5)):
        inv, _ = rotation.rotate_clean_plan(h, w, ang)
        assert np.array_equal(inv, orot.cv_invert_affine(orot.clean_rotation_plan(h, w, ang)['affine'])), (h, w, ang)


@pytest.mark.gpu
@pytest.mark.parametrize('mode,dtype', [('random3Drotate', np.uint16), ('random3Drotate', np.uint8),
                                        ('random90rotate', np.uint16)])
def test_rotated_crop_on_device(mode, dtype):
    import torch
    from neuroclear_amd.data.singlevolume_dataset import SingleVolumeDataset
    rng = np.random.default_rng(5)
    vol = (rng.random((20, 48, 56)) * np.iinfo(dtype).max).astype(dtype)
    opt = Namespace(preprocess=mode + '_randomcrop_addColorChannel_addBatchChannel', crop_size=[12, 16, 20],
                    gpu_ids=[0], isTrain=True, dataroot=None, direction='AtoB')
    ds = SingleVolumeDataset(opt, volume=vol)
    for trial in range(4):
        random.seed(100 + trial)
        np.random.seed(200 + trial)
        got = ds[0]['A']
        assert got.shape == (1, 1, 12, 16, 20)
        # replay the reference's draw order against the oracle
        random.seed(100 + trial)
        np.random.seed(200 + trial)
        angle = random.randint(0, 359) if mode == 'random3Drotate' else int(np.random.choice((-90, 90, -180, 180, -270, 270)))
        rot = orot.rotate_clean_3D_xy(vol, angle)
        z = random.randint(0, rot.shape[0] - 12)
        y = random.randint(0, rot.shape[1] - 16)
        x = random.randint(0, rot.shape[2] - 20)
        ref = (rot[z:z + 12, y:y + 16, x:x + 20] / float(np.iinfo(dtype).max)).astype(np.float32)
        # the same fixed-point arithmetic on both sides (cv2.warpAffine's, restated): bit for bit
        assert np.array_equal(got[0, 0].cpu().numpy(), ref), (angle, float(np.abs(got[0, 0].cpu().numpy() - ref).max() * np.iinfo(dtype).max))
