"""Helpers of the dice path (reference: util/util.py).  `pad_for_dicing` geometry (:196-215) is pure index arithmetic; the
padding itself is never materialised on the MI355X path (the cube cutter treats the pad region as zeros).  The image
metrics of the test script's report (test_dice.py:232-270) -- `normalize` (:56-71), `standardize` (:111-112), `get_mse`,
`get_psnr` (:114-119) -- are float64 numpy reductions over host volumes; they run once per volume, after the device work,
and are pinned bit for bit to values produced by the reference's own functions (tests/golden/postproc_metrics.npz)."""
import math
import os

import numpy as np


def dicing_pad(shape, roi_size, overlap=0):
    """Per-axis zero padding appended by pad_for_dicing: step*((L+ov)//step) + roi - L  (util/util.py:201-211)."""
    step = roi_size - overlap
    return tuple(step * ((int(L) + overlap) // step) + roi_size - int(L) for L in shape)


def padded_shape(shape, roi_size, overlap=0):
    return tuple(int(L) + p for L, p in zip(shape, dicing_pad(shape, roi_size, overlap)))


def pad_for_dicing(image, roi_size, overlap=0):
    """Host (numpy) version with the reference's signature, for callers that want the padded array itself."""
    pz, py, px = dicing_pad(image.shape, roi_size, overlap)
    return np.pad(image, pad_width=((0, pz), (0, py), (0, px)))


def grid_steps(padded, roi_size, overlap):
    """(z_steps, y_steps, x_steps) of data/diceImage_dataset.py:91-93 / util/assemble_dice.py:23-25."""
    step = roi_size - overlap
    return tuple((int(L) - overlap) // step for L in padded)


def mkdir(path):
    os.makedirs(path, exist_ok=True)


def normalize(img_np, data_type=float):
    """Linear map of [min, max] onto [0, 255] / [0, 65535] / [0, 1], cast by truncation (util/util.py:56-71)."""
    lo, hi = np.min(img_np), np.max(img_np)
    if data_type == np.uint8:
        top = 2 ** 8 - 1
    elif data_type == np.uint16:
        top = 2 ** 16 - 1
    elif data_type == float:
        top = 1
    else:
        raise ValueError('normalize: data_type must be np.uint8, np.uint16 or float')
    return ((img_np - lo) * (top / (hi - lo)) + 0).astype(data_type)


def standardize(img_np):
    return (img_np - np.mean(img_np)) / np.std(img_np)


def get_mse(source, target):
    return np.mean((target - source) ** 2)


def get_psnr(source, target, data_range):
    t, s = target.astype(float), source.astype(float)
    return 20 * math.log(data_range, 10) - 10 * math.log(np.mean((t - s) ** 2), 10)


def save_image(image_numpy, image_path):
    """One 2-D image as a .tif (util/util.py:140-154 saves through PIL; here the TIFF is written directly)."""
    from . import tiff
    tiff.imsave(image_path, image_numpy)
