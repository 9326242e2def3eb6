"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the reference's dice / assemble path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

Restates, as plain functions over numpy arrays:

* ``pad_amounts`` / ``pad_for_dicing``  -- /root/reference/util/util.py:196-215
* ``grid_steps``                        -- data/diceImage_dataset.py:91-93  (== util/assemble_dice.py:23-25)
* ``index_to_zyx``                      -- data/diceImage_dataset.py:100-106 (== assemble_dice.py:60-66), x fastest
* ``reflect_pad`` + ``cut_cube``        -- data/diceImage_dataset.py:95-96, 108-120
* ``normalize``                         -- data/base_dataset.py:134-143 (+ float32 cast of :291-295)
* ``assemble``                          -- util/assemble_dice.py:130-213 (crop border, overlap-add /8, count,
                                           *8/count, scale, truncating cast, crop the dicing pad)

Parity pin: tests/golden/dice_*.npz, produced by running the reference's own DiceImageDataSet + Assemble_Dice
(oracle/gen_golden.py).  ``--normalize_intensity`` / ``--histogram_match`` (scikit-image arithmetic) are parity
UNPINNED and are not restated here.
"""
import numpy as np


def pad_amounts(shape, roi, overlap):
    step = roi - overlap
    return tuple(step * ((L + overlap) // step) + roi - L for L in shape)


def pad_for_dicing(vol, roi, overlap):
    pz, py, px = pad_amounts(vol.shape, roi, overlap)
    return np.pad(vol, ((0, pz), (0, py), (0, px)))


def grid_steps(padded_shape, roi, overlap):
    step = roi - overlap
    return tuple((L - overlap) // step for L in padded_shape)


def index_to_zyx(index, steps):
    zs, ys, xs = steps
    return index // (xs * ys), (index % (xs * ys)) // xs, index % xs


def reflect_pad(padded, border):
    return np.pad(padded, ((border, border),) * 3, mode='reflect')


def cut_cube(reflected, index, steps, roi, overlap, border):
    step = roi - overlap
    zi, yi, xi = index_to_zyx(index, steps)
    e = roi + 2 * border
    z0, y0, x0 = zi * step, yi * step, xi * step
    return reflected[z0:z0 + e, y0:y0 + e, x0:x0 + e]


def normalize(cube):
    if cube.dtype == np.uint8:
        den = 2 ** 8 * 1.0 - 1
    elif cube.dtype == np.uint16:
        den = 2 ** 16 * 1.0 - 1
    else:
        raise TypeError('dice input must be uint8/uint16 (reference leaves other dtypes unbound)')
    return (cube / den).astype(float).astype(np.float32)


def percentile_np121(a, q):
    """np.percentile(a, q) (default linear method) of a float32 array with the arithmetic of numpy 1.21.2, the version
    the reference pins (conda_environment/neuroclear_env.yml:155; numpy/lib/function_base.py _quantile_ureduce_func /
    _lerp): index q/100 * (n-1) in float64, neighbours below / above, weight t; lerp = a + (b-a)*t, replaced by
    b - (b-a)*(1-t) where t >= 0.5; (b - a) is a float32 subtraction, everything else float64.  (numpy >= 2 keeps
    float32 throughout, which is why this is a restatement and not a call.)  PARITY UNPINNED against 1.21.2 itself;
    tests pin the selection / interpolation logic against the installed numpy on float64 data, where both agree."""
    s = np.sort(np.asarray(a).ravel())
    n = s.size
    h = (q / 100.0) * (n - 1)
    lo = int(np.floor(h))
    hi = min(lo + 1, n - 1)
    t = h - lo
    A, B = s[lo], s[hi]
    d = np.float64(B - A)  # B - A in the array's own precision
    return float(np.float64(B) - d * (1.0 - t)) if t >= 0.5 else float(np.float64(A) + d * t)


def rescale_intensity_f32(image, imin, imax):
    """skimage.exposure.rescale_intensity(image, in_range=(imin, imax)) for a float32 image, restated from scikit-image
    0.18.3 (skimage/exposure/exposure.py; the version pinned at neuroclear_env.yml:203): out_range 'dtype' of a float
    image is (0, 1) when imin >= 0, else (-1, 1); clip, (image - imin) / (imax - imin), * (omax - omin) + omin, with the
    python-float scalars cast to the array's float32 (numpy 1.x value-based casting).  PARITY UNPINNED (not installed)."""
    imin, imax = float(imin), float(imax)
    omin, omax = (0.0, 1.0) if imin >= 0 else (-1.0, 1.0)
    image = np.clip(image, np.float32(imin), np.float32(imax))
    image = (image - np.float32(imin)) / np.float32(imax - imin)
    return (image * np.float32(omax - omin) + np.float32(omin)).astype(np.float32)


def assemble(cubes, padded_shape, original_shape, roi, overlap, border, data_type='uint16', normalize=None):
    """cubes: iterable of (R+2b)^3 float32 arrays in index order.  Returns the cropped uint8/uint16 volume.
    normalize = (p_lo, p_hi) percent: --normalize_intensity with --sat_level (util/assemble_dice.py:188-192)."""
    if border < 1:
        raise ValueError('border_cut must be >= 1 (reference slices [b:-b]; b=0 yields an empty cube)')
    steps = grid_steps(padded_shape, roi, overlap)
    step = roi - overlap
    acc = np.zeros(padded_shape, dtype=np.float32)
    cnt = np.zeros(padded_shape, dtype=np.float32)
    n = 0
    for index, cube in enumerate(cubes):
        c = np.asarray(cube, dtype=np.float32)[border:-border, border:-border, border:-border]
        assert c.shape == (roi, roi, roi)
        zi, yi, xi = index_to_zyx(index, steps)
        z0, y0, x0 = zi * step, yi * step, xi * step
        if overlap > 0:
            acc[z0:z0 + roi, y0:y0 + roi, x0:x0 + roi] += c / 8
            cnt[z0:z0 + roi, y0:y0 + roi, x0:x0 + roi] += np.ones((roi, roi, roi), dtype=np.float32)
        n += 1
    assert n == steps[0] * steps[1] * steps[2]
    return _finish(acc, cnt, padded_shape, original_shape, overlap, data_type, normalize)


def finalize(acc, padded_shape, original_shape, roi, overlap, data_type='uint16', normalize=None):
    """The tail of assemble_all (util/assemble_dice.py:176-213) on an accumulator that already holds the sum of cube / 8
    over all cubes (e.g. the reduce(sum) of per-rank accumulators): count volume, (acc / count) * 8, scale, truncating
    cast, crop of the dicing pad."""
    steps = grid_steps(padded_shape, roi, overlap)
    step = roi - overlap
    cnt = np.zeros(padded_shape, dtype=np.float32)
    for index in range(steps[0] * steps[1] * steps[2]):
        zi, yi, xi = index_to_zyx(index, steps)
        cnt[zi * step:zi * step + roi, yi * step:yi * step + roi, xi * step:xi * step + roi] += 1.0
    return _finish(np.array(acc, dtype=np.float32), cnt, padded_shape, original_shape, overlap, data_type, normalize)


def _finish(acc, cnt, padded_shape, original_shape, overlap, data_type, normalize):
    if overlap > 0:
        acc = (acc / cnt) * 8
    if normalize is not None:
        p1_, p99_ = percentile_np121(acc, normalize[0]), percentile_np121(acc, normalize[1])
        acc = rescale_intensity_f32(acc, p1_, p99_)
    if data_type == 'uint8':
        acc *= 255
        acc = acc.astype(np.uint8)
    elif data_type == 'uint16':
        acc *= 2 ** 16 - 1
        acc = acc.astype(np.uint16)
    pads = [padded_shape[i] - original_shape[i] for i in range(3)]
    return acc[:-pads[0], :-pads[1], :-pads[2]]


def match_histograms_np(source, template):
    """skimage.exposure.match_histograms(image, reference) for single-channel arrays, restated from scikit-image 0.18.3
    (exposure/histogram_matching.py::_match_cumulative_cdf) -- the reference calls it per cube at
    util/assemble_dice.py:150-151.  PARITY UNPINNED against scikit-image itself (not installed here); the two numpy
    functions it is made of (np.unique, np.interp) are the real ones.  Returns float64, like scikit-image."""
    src_values, src_unique_indices, src_counts = np.unique(source.ravel(), return_inverse=True, return_counts=True)
    tmpl_values, tmpl_counts = np.unique(template.ravel(), return_counts=True)
    src_quantiles = np.cumsum(src_counts) / source.size
    tmpl_quantiles = np.cumsum(tmpl_counts) / template.size
    interp_a_values = np.interp(src_quantiles, tmpl_quantiles, tmpl_values)
    return interp_a_values[src_unique_indices.ravel()].reshape(source.shape)
