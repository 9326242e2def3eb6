"""--normalize_intensity.  CPU: the oracle's restatement of np.percentile against the installed
numpy on float64 data (where numpy 1.21 and 2.x agree).  GPU: exact radix select / percentile against the oracle, and
the assembler with --normalize_intensity against the oracle's assembler bit for bit."""
from argparse import Namespace

import numpy as np
import pytest

from oracle import dice as odice


@pytest.mark.parametrize('seed', range(6))
def test_oracle_percentile_matches_numpy_on_float64(seed):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(int(rng.integers(10, 4000)))
    if seed % 2:
        a = np.round(a, 1)  # many duplicates
    for q in (0.0, 0.25, 1.0, 33.3, 50.0, 99.0, 99.75, 100.0):
        assert odice.percentile_np121(a, q) == float(np.percentile(a, q)), (seed, q)


@pytest.mark.gpu
def test_radix_select_and_percentile():
    import torch
    from neuroclear_amd.util import percentile as pct
    rng = np.random.default_rng(3)
    for n, kind in ((1000, 'u'), (70001, 'n'), (300000, 'dup'), (5, 'u')):
        a = rng.random(n) if kind == 'u' else rng.standard_normal(n)
        if kind == 'dup':
            a = np.round(a, 2)
        a = a.astype(np.float32)
        x = torch.from_numpy(a).cuda()
        s = np.sort(a)
        for k in (0, n // 3, n // 2, n - 2 if n > 1 else 0, n - 1):
            assert pct.select_kth(x, k) == float(s[k]), (n, kind, k)
        qs = (0.25, 1.0, 50.0, 99.75)
        got = pct.percentile(x, qs)
        for q, g in zip(qs, got):
            assert g == odice.percentile_np121(a, q), (n, kind, q)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', ['uint16', 'uint8'])
def test_assemble_normalize_intensity(dtype):
    import torch
    from neuroclear_amd.util.assemble_dice import Assemble_Dice
    from neuroclear_amd.util import util
    R, ov, b, L = 32, 4, 4, (70, 50, 61)
    opt = Namespace(dice_size=[R] * 3, overlap=ov, border_cut=b, gpu_ids=[0], skip_real=True, data_type=dtype,
                    histogram_match=False, normalize_intensity=True, sat_level=[0.25, 99.75])
    asm = Assemble_Dice(opt, L)
    padded = util.padded_shape(L, R, ov)
    n = asm.len_cube_queue
    E = R + 2 * b
    rng = np.random.default_rng(9)
    cubes = [(rng.random((E, E, E)) ** 2).astype(np.float32) for _ in range(n)]
    for c in cubes:
        asm.addToStack(dict(fake=torch.from_numpy(c)[None, None].cuda()))
    asm.assemble_all()
    got = asm.getDict()['fake']
    ref = odice.assemble(cubes, padded, L, R, ov, b, dtype, normalize=(0.25, 99.75))
    assert got.dtype == ref.dtype and got.shape == ref.shape
    d = np.abs(got.astype(np.int64) - ref.astype(np.int64))
    assert int(d.max()) == 0, (int(d.max()), float((d > 0).mean()))
