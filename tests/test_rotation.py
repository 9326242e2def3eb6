"""Rotation augmentation.  CPU: the product's geometry (neuroclear_amd/data/rotation.py) against
the oracle's restatement (oracle/rotate.py) and closed-form cases.  GPU: nc_rotate_crop through SingleVolumeDataset
against the oracle's numpy warp -- the interpolation itself is parity-UNPINNED against OpenCV (not installed here), see
the headers of both modules."""
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
        err = np.abs(got[0, 0].cpu().numpy() - ref) * np.iinfo(dtype).max
        # identical arithmetic (fp64 bilinear, round-half-even) on both sides: at most a tie broken differently
        assert err.max() <= 1.0 + 1e-3, (angle, err.max())
        assert (err > 1e-3).mean() < 1e-3
