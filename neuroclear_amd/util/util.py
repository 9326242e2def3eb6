"""Geometry helpers of the dice path (reference: util/util.py:196-215 `pad_for_dicing`).  Pure index arithmetic; the
padding itself is never materialised on the MI355X path (the cube cutter treats the pad region as zeros)."""
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
