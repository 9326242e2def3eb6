"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's "clean rotation" augmentation.

Only ``tests/`` may import this module.  PARITY UNPINNED for the interpolation: the reference rotates every z-slice with
``cv2.warpAffine(..., flags=cv2.INTER_LINEAR)`` (data/base_dataset.py:364-370) and OpenCV is not installed in the build
container, so no golden vector could be generated.  What is restated line by line from the reference is the GEOMETRY --

* ``rotate_image``            data/base_dataset.py:306-372  (cv2.getRotationMatrix2D formula, bounding box of the rotated
                              corners, integer translation, forward affine matrix);
* ``largest_rotated_rect``    :375-408  (including its ``gamma = atan2(bb_w, bb_w)`` in both branches);
* ``crop_around_center``      :411-432;
* ``__rotate_clean`` / ``__rotate_clean_3D_xy`` / ``__randomrotate_clean_3D_xy``  :434-460;

-- and OpenCV's documented semantics of warpAffine without WARP_INVERSE_MAP: dst(x, y) = src(M^-1 (x, y)), bilinear,
BORDER_CONSTANT 0, result rounded to the source integer type.  OpenCV evaluates the bilinear weights in 5-bit fixed point
(INTER_BITS); this restatement (and the HIP kernel it checks) uses exact floating-point weights, so values may differ
from the reference's by a few LSB of uint16 -- the crop geometry, i.e. WHICH voxels are read, is identical.
"""
import math

import numpy as np


def rotation_matrix_2d(center, angle_deg, scale=1.0):
    """cv2.getRotationMatrix2D (documented formula)."""
    a = math.radians(angle_deg)
    alpha, beta = scale * math.cos(a), scale * math.sin(a)
    cx, cy = center
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]])


def rotate_geometry(h, w, angle_deg):
    """rotate_image (base_dataset.py:306-372) up to the warp: returns (new_w, new_h, affine 2x3 src -> dst)."""
    image_size = (w, h)
    image_center = tuple(np.array(image_size) / 2)
    rot_mat = np.vstack([rotation_matrix_2d(image_center, angle_deg, 1.0), [0, 0, 1]])
    rot_nt = rot_mat[0:2, 0:2]  # (row vector) * matrix, as the reference's np.matrix product
    w2, h2 = image_size[0] * 0.5, image_size[1] * 0.5
    rc = [np.array([-w2, h2]) @ rot_nt, np.array([w2, h2]) @ rot_nt,
          np.array([-w2, -h2]) @ rot_nt, np.array([w2, -h2]) @ rot_nt]
    xs = [p[0] for p in rc]
    ys = [p[1] for p in rc]
    right, left = max(x for x in xs if x > 0), min(x for x in xs if x < 0)
    top, bot = max(y for y in ys if y > 0), min(y for y in ys if y < 0)
    new_w, new_h = int(abs(right - left)), int(abs(top - bot))
    trans = np.array([[1, 0, int(new_w * 0.5 - w2)], [0, 1, int(new_h * 0.5 - h2)], [0, 0, 1.0]])
    affine = (trans @ rot_mat)[0:2, :]
    return new_w, new_h, affine


def largest_rotated_rect(w, h, angle):
    """base_dataset.py:375-408, verbatim arithmetic (angle in radians)."""
    quadrant = int(math.floor(angle / (math.pi / 2))) & 3
    sign_alpha = angle if ((quadrant & 1) == 0) else math.pi - angle
    alpha = (sign_alpha % math.pi + math.pi) % math.pi
    bb_w = w * math.cos(alpha) + h * math.sin(alpha)
    bb_h = w * math.sin(alpha) + h * math.cos(alpha)
    gamma = math.atan2(bb_w, bb_w)
    delta = math.pi - alpha - gamma
    length = h if (w < h) else w
    d = length * math.cos(alpha)
    a = d * math.sin(alpha) / math.sin(delta)
    y = a * math.cos(gamma)
    x = y * math.tan(gamma)
    return bb_w - 2 * x, bb_h - 2 * y


def crop_rect(rot_w, rot_h, width, height):
    """crop_around_center (base_dataset.py:411-432): returns (x1, y1, x2, y2) inside the rotated image."""
    cx, cy = int(rot_w * 0.5), int(rot_h * 0.5)
    if width > rot_w:
        width = rot_w
    if height > rot_h:
        height = rot_h
    return int(cx - width * 0.5), int(cy - height * 0.5), int(cx + width * 0.5), int(cy + height * 0.5)


def clean_rotation_plan(h, w, angle_deg):
    """Everything the product needs: inverse affine (dst -> src, 2x3) of the FULL rotated image, and the crop
    rectangle of __rotate_clean; the cleaned slice is rotated[y1:y2, x1:x2]."""
    new_w, new_h, affine = rotate_geometry(h, w, angle_deg)
    inv = np.linalg.inv(np.vstack([affine, [0, 0, 1]]))[0:2, :]
    rw, rh = largest_rotated_rect(w, h, math.radians(angle_deg))
    x1, y1, x2, y2 = crop_rect(new_w, new_h, rw, rh)
    x1, y1 = max(x1, 0), max(y1, 0)  # numpy slicing semantics of image[y1:y2, x1:x2]
    x2, y2 = min(x2, new_w), min(y2, new_h)
    return dict(new_w=new_w, new_h=new_h, inv=inv, rect=(x1, y1, x2, y2))


def warp_bilinear(src, inv, x0, y0, out_h, out_w):
    """dst[y, x] = bilinear(src, inv @ (x + x0, y + y0, 1)), zero outside, rounded to src's integer type."""
    ys, xs = np.meshgrid(np.arange(out_h) + y0, np.arange(out_w) + x0, indexing='ij')
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    fx, fy = np.floor(sx), np.floor(sy)
    ax, ay = sx - fx, sy - fy
    ix, iy = fx.astype(np.int64), fy.astype(np.int64)
    h, w = src.shape

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return np.where(ok, src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.float64), 0.0)
    v = (tap(iy, ix) * (1 - ax) + tap(iy, ix + 1) * ax) * (1 - ay) + (tap(iy + 1, ix) * (1 - ax) + tap(iy + 1, ix + 1) * ax) * ay
    info = np.iinfo(src.dtype)
    return np.clip(np.rint(v), info.min, info.max).astype(src.dtype)


def rotate_clean_3D_xy(vol, angle_deg):
    """__rotate_clean_3D_xy (base_dataset.py:447-453): every z-slice rotated and cropped to the inscribed rectangle."""
    plan = clean_rotation_plan(vol.shape[1], vol.shape[2], angle_deg)
    x1, y1, x2, y2 = plan['rect']
    return np.stack([warp_bilinear(s, plan['inv'], x1, y1, y2 - y1, x2 - x1) for s in vol])
