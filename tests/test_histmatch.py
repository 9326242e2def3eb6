"""--histogram_match (reference util/assemble_dice.py:149-151): oracle restatement of scikit-image 0.18.3's
match_histograms (parity unpinned against scikit-image, which is not installed) and the device implementation."""
import numpy as np
import pytest

from oracle import dice as odice


def _hand_match(src, tmpl):
    """Independent pure-Python statement of the same definition (tiny inputs): out(v) = interp(#{src <= v} / n)."""
    s, t = src.ravel(), tmpl.ravel()
    tv = sorted(set(t.tolist()))
    tq = [np.sum(t <= v) / t.size for v in tv]
    out = []
    for v in s.tolist():
        q = np.sum(s <= v) / s.size
        out.append(np.interp(q, tq, tv))
    return np.array(out).reshape(src.shape)


def test_oracle_match_histograms_definition():
    rng = np.random.default_rng(0)
    src = rng.random((5, 6, 7)).astype(np.float32)
    src[0, :3] = src[1, :3]  # ties
    tmpl = (rng.integers(0, 50, (5, 6, 7)) / np.float32(65535)).astype(np.float32)
    got = odice.match_histograms_np(src, tmpl)
    assert got.dtype == np.float64 and got.shape == src.shape
    np.testing.assert_array_equal(got, _hand_match(src, tmpl))
    # properties of the definition: monotone in the source, values inside the template's range, the largest source
    # value maps to the largest template value
    o = np.argsort(src.ravel(), kind='stable')
    assert np.all(np.diff(got.ravel()[o]) >= 0)
    assert got.min() >= tmpl.min() and got.max() == tmpl.max()


def test_oracle_match_identity():
    """Matching an array to itself is the identity (every quantile sits exactly on a knot)."""
    rng = np.random.default_rng(1)
    a = rng.random(1000).astype(np.float32)
    np.testing.assert_array_equal(odice.match_histograms_np(a, a), a.astype(np.float64))


@pytest.mark.gpu
@pytest.mark.parametrize('n,levels', [(40 ** 3, 65536), (12345, 300), (2048, 4), (1, 1), (120 ** 3, 65536)])
def test_gpu_match_histograms_equals_oracle(n, levels):
    import torch
    from neuroclear_amd.util.assemble_dice import match_histograms
    rng = np.random.default_rng(n)
    src = (1.0 / (1.0 + np.exp(-rng.normal(0, 2, n)))).astype(np.float32)  # sigmoid outputs, like the generator's
    src[: n // 7] = src[n // 7: 2 * (n // 7)]                                # with ties
    if n > 10:
        src[5] = -0.0
        src[6] = 0.0
    tmpl = (rng.integers(0, levels, n) / np.float32(65535)).astype(np.float32)
    want = odice.match_histograms_np(src, tmpl).astype(np.float32)
    got = match_histograms(torch.from_numpy(src).cuda(), torch.from_numpy(tmpl).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    # generic float template (not on an integer grid)
    tmpl2 = rng.random(n).astype(np.float32)
    want2 = odice.match_histograms_np(src, tmpl2).astype(np.float32)
    got2 = match_histograms(torch.from_numpy(src).cuda(), torch.from_numpy(tmpl2).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got2, want2)


@pytest.mark.gpu
def test_gpu_diced_inference_with_histogram_match():
    """End to end: test_dice's loop with --histogram_match vs the oracle pipeline (oracle assemble of oracle-matched
    cubes produced from the SAME device network outputs): <= 1 LSB (the reference adds a float64 cube into a float32
    stack, this path rounds the matched cube to float32 first)."""
    import torch
    from argparse import Namespace
    from neuroclear_amd.data.diceImage_dataset import DiceImageDataSet
    from neuroclear_amd.test_dice import diced_inference
    rng = np.random.default_rng(3)
    vol = rng.integers(0, 65536, (40, 40, 40), dtype=np.uint16)
    opt = Namespace(dice_size=[16] * 3, overlap=4, border_cut=2, gpu_ids=[0], skip_real=True, data_type='uint16',
                    histogram_match=True, normalize_intensity=False)

    class Net(torch.nn.Module):  # a smooth non-linear stand-in for the generator
        def forward(self, x):
            return torch.sigmoid(3 * x - 1 + 0.3 * torch.roll(x, 1, -1))

    net = Net().cuda()
    out = diced_inference(net, vol, opt)
    # oracle side: same cubes, same network outputs (taken from the device), matching + assembly on the CPU
    ds = DiceImageDataSet(opt, volume=vol)
    E, b = 20, 2
    fakes = []
    with torch.no_grad():
        for i in range(len(ds)):
            x = ds[i]['A'].unsqueeze(0)
            y = net(x).reshape(E, E, E).cpu().numpy()
            xr = x.reshape(E, E, E).cpu().numpy()
            m = y.copy()
            m[b:-b, b:-b, b:-b] = odice.match_histograms_np(y[b:-b, b:-b, b:-b], xr[b:-b, b:-b, b:-b]).astype(np.float32)
            fakes.append(m)
    want = odice.assemble(fakes, tuple(s + p for s, p in zip(vol.shape, odice.pad_amounts(vol.shape, 16, 4))), vol.shape, 16, 4, 2, 'uint16')
    assert out.shape == want.shape and out.dtype == want.dtype
    assert np.abs(out.astype(np.int64) - want.astype(np.int64)).max() <= 1
