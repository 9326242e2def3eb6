"""Geometry of the reference's "clean rotation" augmentation (data/base_dataset.py:306-460), host side.

`rotate_image` (:306-372) builds the forward affine of a rotation about the slice centre into an enlarged canvas,
`largest_rotated_rect` (:375-408) + `crop_around_center` (:411-432) cut the inscribed axis-aligned rectangle out of it.
The reference then warps every z-slice of the WHOLE volume with cv2 on the host; here this module only produces the
numbers -- canvas size, inverse affine, rectangle -- and nc_rotate_crop (HIP) samples just the voxels of the training
crop.  Same names and arithmetic as the reference so the two can be read side by side.  Interpolation note: OpenCV's
INTER_LINEAR uses 5-bit fixed-point weights, the kernel exact bilinear weights (values can differ by a few LSB of
uint16; which voxels are read is identical) -- OpenCV is not installed in the build container, parity unpinned."""
import math

import numpy as np


def getRotationMatrix2D(center, angle, scale):
    a = math.radians(angle)
    alpha, beta = scale * math.cos(a), scale * math.sin(a)
    return np.array([[alpha, beta, (1 - alpha) * center[0] - beta * center[1]],
                     [-beta, alpha, beta * center[0] + (1 - alpha) * center[1]]])


def rotate_image_geometry(h, w, angle):
    """rotate_image (:306-372) without the warp: (new_w, new_h, affine_mat 2x3 source -> canvas)."""
    image_size = (w, h)
    image_center = tuple(np.array(image_size) / 2)
    rot_mat = np.vstack([getRotationMatrix2D(image_center, angle, 1.0), [0, 0, 1]])
    r = rot_mat[0:2, 0:2]
    image_w2, image_h2 = image_size[0] * 0.5, image_size[1] * 0.5
    corners = [np.array([-image_w2, image_h2]) @ r, np.array([image_w2, image_h2]) @ r,
               np.array([-image_w2, -image_h2]) @ r, np.array([image_w2, -image_h2]) @ r]
    x_coords = [pt[0] for pt in corners]
    y_coords = [pt[1] for pt in corners]
    right_bound = max(x for x in x_coords if x > 0)
    left_bound = min(x for x in x_coords if x < 0)
    top_bound = max(y for y in y_coords if y > 0)
    bot_bound = min(y for y in y_coords if y < 0)
    new_w = int(abs(right_bound - left_bound))
    new_h = int(abs(top_bound - bot_bound))
    trans_mat = np.array([[1, 0, int(new_w * 0.5 - image_w2)], [0, 1, int(new_h * 0.5 - image_h2)], [0, 0, 1.0]])
    return new_w, new_h, (trans_mat @ rot_mat)[0:2, :]


def largest_rotated_rect(w, h, angle):
    """:375-408 (angle in radians); gamma is atan2(bb_w, bb_w) in both branches, as in the reference."""
    quadrant = int(math.floor(angle / (math.pi / 2))) & 3
    sign_alpha = angle if ((quadrant & 1) == 0) else math.pi - angle
    alpha = (sign_alpha % math.pi + math.pi) % math.pi
    bb_w = w * math.cos(alpha) + h * math.sin(alpha)
    bb_h = w * math.sin(alpha) + h * math.cos(alpha)
    gamma = math.atan2(bb_w, bb_w) if (w < h) else math.atan2(bb_w, bb_w)
    delta = math.pi - alpha - gamma
    length = h if (w < h) else w
    d = length * math.cos(alpha)
    a = d * math.sin(alpha) / math.sin(delta)
    y = a * math.cos(gamma)
    x = y * math.tan(gamma)
    return bb_w - 2 * x, bb_h - 2 * y


def crop_around_center_rect(image_w, image_h, width, height):
    """crop_around_center (:411-432) as a rectangle (x1, y1, x2, y2), clipped like numpy slicing clips."""
    image_center = (int(image_w * 0.5), int(image_h * 0.5))
    if width > image_w:
        width = image_w
    if height > image_h:
        height = image_h
    x1 = int(image_center[0] - width * 0.5)
    x2 = int(image_center[0] + width * 0.5)
    y1 = int(image_center[1] - height * 0.5)
    y2 = int(image_center[1] + height * 0.5)
    return max(x1, 0), max(y1, 0), min(x2, image_w), min(y2, image_h)


def rotate_clean_plan(image_height, image_width, angle):
    """__rotate_clean (:434-445): inverse affine (canvas pixel -> source pixel) and the rectangle kept."""
    new_w, new_h, affine = rotate_image_geometry(image_height, image_width, angle)
    inv = np.linalg.inv(np.vstack([affine, [0, 0, 1]]))[0:2, :]
    rw, rh = largest_rotated_rect(image_width, image_height, math.radians(angle))
    return np.ascontiguousarray(inv, dtype=np.float64), crop_around_center_rect(new_w, new_h, rw, rh)
