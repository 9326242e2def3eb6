"""The tail of the test script (reference test_dice.py:126-270): TIFF container, image metrics.
CPU: `normalize` / `standardize` / `get_psnr` bit for bit against values produced by the reference's own util/util.py
(tests/golden/postproc_metrics.npz, oracle/gen_golden.py::gen_postproc); the TIFF writer against PIL's reader and its own
reader (also BigTIFF-free large-ish stacks, both byte orders on read).  GPU: the test script end to end with every output
switch on a small volume."""
import os

import numpy as np
import pytest

from neuroclear_amd.util import tiff
from neuroclear_amd.util import util as U


def test_metrics_match_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, 'postproc_metrics.npz'))
    rng = np.random.default_rng(int(g['seed']))
    real = rng.integers(0, 65536, (24, 30, 36), dtype=np.uint16)
    fake = np.clip(real.astype(np.float64) * 0.8 + rng.normal(0, 900, real.shape), 0, 65535).astype(np.uint16)
    gt = np.clip(real.astype(np.float64) * 0.9 + rng.normal(0, 300, real.shape) + 500, 0, 65535).astype(np.uint16)
    n = {}
    for k, v in (('real', real), ('fake', fake), ('gt', gt)):
        for _ in range(2):
            v = U.normalize(U.standardize(v), data_type=np.uint8)
        n[k] = v
        assert v.dtype == np.uint8 and np.array_equal(v, g['n_' + k]), k
    assert np.array_equal(U.standardize(real), g['std_real'])
    assert np.array_equal(U.normalize(fake.astype(np.float64), data_type=np.uint16), g['norm16_fake'])
    assert U.get_psnr(n['real'], n['gt'], 255) == float(g['psnr_in'])
    assert U.get_psnr(n['fake'], n['gt'], 255) == float(g['psnr_out'])
    assert U.get_mse(n['fake'].astype(float), n['gt'].astype(float)) == float(g['mse'])


@pytest.mark.parametrize('dtype', [np.uint8, np.uint16, np.float32])
def test_tiff_roundtrip_and_pil(tmp_path, dtype):
    from PIL import Image
    a = (np.random.default_rng(0).random((6, 19, 31)) * 250).astype(dtype)
    p = str(tmp_path / 'v.tif')
    tiff.imsave(p, a)
    b = tiff.imread(p)
    assert b.dtype == a.dtype and np.array_equal(a, b)
    im = Image.open(p)
    assert im.n_frames == 6
    for i in range(6):
        im.seek(i)
        assert np.array_equal(np.array(im), a[i])
    tiff.imsave(p, a[2])
    assert np.array_equal(tiff.imread(p), a[2])
    # a file written by another program (PIL), big-endian header included
    Image.fromarray(a[1]).save(str(tmp_path / 'p.tif'))
    assert np.array_equal(tiff.imread(str(tmp_path / 'p.tif')), a[1])
    with pytest.raises(TypeError):
        tiff.imsave(p, a.astype(np.float64))


def test_tiff_reads_big_endian_and_bigtiff(tmp_path):
    import struct
    a = np.arange(12, dtype=np.uint16).reshape(3, 4) * 1000
    # hand-built big-endian classic TIFF, one strip
    data = a.astype('>u2').tobytes()
    tags = [(256, 3, 4), (257, 3, 3), (258, 3, 16), (259, 3, 1), (262, 3, 1), (273, 4, 8), (277, 3, 1), (278, 3, 3), (279, 4, len(data))]
    ifd_off = 8 + len(data)
    buf = b'MM' + struct.pack('>HI', 42, ifd_off) + data + struct.pack('>H', len(tags))
    for t, ty, v in tags:
        buf += struct.pack('>HHI', t, ty, 1) + (struct.pack('>HH', v, 0) if ty == 3 else struct.pack('>I', v))
    buf += struct.pack('>I', 0)
    p = str(tmp_path / 'be.tif')
    open(p, 'wb').write(buf)
    assert np.array_equal(tiff.imread(p), a)


@pytest.mark.gpu
def test_test_dice_script_outputs(tmp_path):
    """neuroclear_amd.test_dice.main end to end: checkpoint -> diced inference -> volume TIFF, projections, slices, PSNR."""
    import torch
    from neuroclear_amd import test_dice
    from neuroclear_amd.models import networks
    from neuroclear_amd.util import seed as S
    d, dg, ck = tmp_path / 'data', tmp_path / 'gt', tmp_path / 'ck' / 'exp'
    for q in (d, dg, ck):
        q.mkdir(parents=True)
    vol = S.random_volume(5, (40, 44, 48))
    tiff.imsave(str(d / 'vol.tif'), vol)           # the volume goes in as a TIFF stack
    tiff.imsave(str(dg / 'gt.tif'), (vol // 2 + 100).astype(np.uint16))
    net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
    torch.save({k: v.cpu() for k, v in net.state_dict().items()}, str(ck / 'latest_net_G.pth'))
    m = test_dice.main(['--dataroot', str(d), '--dataroot_gt', str(dg), '--name', 'exp', '--checkpoints_dir', str(tmp_path / 'ck'),
                        '--results_dir', str(tmp_path / 'res'), '--dice_size', '24', '24', '24', '--overlap', '4', '--border_cut', '4',
                        '--no_dropout', '--save_volume', '--save_projections', '--save_slices', '--init_type', 'kaiming'])
    web = tmp_path / 'res' / 'exp' / 'test_latest'
    out = tiff.imread(str(web / 'volumes' / 'output_volume_xy-view_epoch-latest.tif'))
    real = tiff.imread(str(web / 'volumes' / 'input_volume_xy-view.tif'))
    assert out.shape == vol.shape and out.dtype == np.uint16
    assert int(np.abs(real.astype(np.int64) - vol).max()) <= 1   # dice -> assemble of the input: identity up to 1 LSB
    assert np.array_equal(np.load(str(web / 'volumes' / 'output_volume.npy')), out)
    assert np.array_equal(tiff.imread(str(web / 'projections' / 'fake_xy_proj_epoch-latest.tif')), out.max(0))
    assert np.array_equal(tiff.imread(str(web / 'projections' / 'real_xz_proj.tif')), real.max(1))
    assert np.array_equal(tiff.imread(str(web / 'images' / 'output_xz' / 'output_xz_7.tif')), out[:, 7, :])
    assert np.array_equal(tiff.imread(str(web / 'images' / 'input_yz' / 'input_yz_47.tif')), real[:, :, 47])
    assert len(os.listdir(str(web / 'images' / 'output_xy'))) == 40
    txt = open(str(web / 'metrics.txt')).read()
    assert 'Network Output vs. Groundtruth' in txt and ('(psnr: %.4f)' % m['psnr_output_gt']) in txt
    assert np.isfinite(m['psnr_input_gt']) and m['psnr_input_gt'] > m['psnr_output_gt']  # a random network is worse than the input
