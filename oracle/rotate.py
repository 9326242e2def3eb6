"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's "clean rotation" augmentation.

Only ``tests/`` may import this module.  The reference rotates every z-slice with ``cv2.warpAffine(..., flags=cv2.INTER_LINEAR)``
(data/base_dataset.py:364-370); OpenCV (pinned: opencv 4.5.0, conda_environment/neuroclear_env.yml:159) is a third-party dependency that is
absent here and cannot be installed, so the interpolation is RESTATED FROM THE PUBLISHED ALGORITHM (``warp_affine_cv`` below: OpenCV
4.5.0 modules/imgproc/src/imgwarp.cpp, cv::warpAffine -> hal::warpAffine / WarpAffineInvoker -> cv::remap / remapBilinear, and
initInterTab2D for the weight tables) and anchored on known-answer cases that follow from that algorithm by hand (tests/test_rotation.py);
no golden vector from cv2 itself could be generated.  What is restated line by line from the reference is the GEOMETRY --

* ``rotate_image``            data/base_dataset.py:306-372  (cv2.getRotationMatrix2D formula, bounding box of the rotated
                              corners, integer translation, forward affine matrix);
* ``largest_rotated_rect``    :375-408  (including its ``gamma = atan2(bb_w, bb_w)`` in both branches);
* ``crop_around_center``      :411-432;
* ``__rotate_clean`` / ``__rotate_clean_3D_xy`` / ``__randomrotate_clean_3D_xy``  :434-460;

-- and OpenCV's warpAffine without WARP_INVERSE_MAP, INTER_LINEAR, BORDER_CONSTANT 0 (``warp_affine_cv``): the forward matrix is
inverted in double precision with OpenCV's own formula; source coordinates are FIXED POINT -- AB_BITS = 10 bits for the row / column
terms (each rounded half-to-even: cvRound), a round_delta of 2^(10-5-1), then INTER_BITS = 5 fractional bits -- so the bilinear
weights are multiples of 1/32 per axis; 8-bit images blend with the 15-bit integer table (INTER_REMAP_COEF_BITS) and a rounding
shift, 16-bit images with the float table in float32 arithmetic (products and sums left to right, then cvRound and saturation).
``warp_bilinear`` (exact weights in float64) is kept as the textbook form the fixed-point result is compared with in the tests.
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
    return dict(new_w=new_w, new_h=new_h, inv=inv, affine=affine, rect=(x1, y1, x2, y2))


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


def cv_invert_affine(M):
    """cv::warpAffine's inversion of the 2x3 forward matrix (imgwarp.cpp, the `!(flags & WARP_INVERSE_MAP)` block), operation
    for operation in float64."""
    M = [float(v) for v in np.asarray(M, np.float64).ravel()]
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return np.array(M, np.float64).reshape(2, 3)


AB_BITS, INTER_BITS = 10, 5                       # imgwarp.cpp: AB_BITS = MAX(10, INTER_BITS); INTER_TAB_SIZE = 32
AB_SCALE, INTER_TAB_SIZE = 1 << AB_BITS, 1 << INTER_BITS
INTER_REMAP_COEF_BITS = 15


def cv_fixed_coords(Minv, xs, ys):
    """WarpAffineInvoker (INTER_LINEAR branch): integer source coordinates and 5-bit fractions of destination pixels (xs, ys).
    adelta / bdelta = cvRound(M[0] x AB_SCALE), cvRound(M[3] x AB_SCALE); X0 / Y0 = cvRound((M[1] y + M[2]) AB_SCALE) + round_delta, ...;
    np.rint is round-half-to-even like cvRound(double)."""
    m = np.asarray(Minv, np.float64)
    xs = np.asarray(xs, np.float64)
    ys = np.asarray(ys, np.float64)
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    adelta = np.rint(m[0, 0] * xs * AB_SCALE).astype(np.int64)
    bdelta = np.rint(m[1, 0] * xs * AB_SCALE).astype(np.int64)
    X0 = np.rint((m[0, 1] * ys + m[0, 2]) * AB_SCALE).astype(np.int64) + round_delta
    Y0 = np.rint((m[1, 1] * ys + m[1, 2]) * AB_SCALE).astype(np.int64) + round_delta
    X = (X0 + adelta) >> (AB_BITS - INTER_BITS)
    Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS)
    return X >> INTER_BITS, Y >> INTER_BITS, X & (INTER_TAB_SIZE - 1), Y & (INTER_TAB_SIZE - 1)


def warp_affine_cv(src, M_fwd, x0, y0, out_h, out_w):
    """Rows y0 .. y0 + out_h - 1, columns x0 .. x0 + out_w - 1 of cv2.warpAffine(src, M_fwd, dsize, flags=INTER_LINEAR) for a 2-D uint8 or
    uint16 image (the values do not depend on dsize).  remapBilinear: taps outside the image are the border value 0; uint8:
    (sum of v * itab + 2^14) >> 15 with itab = 32 (32 - fy)(32 - fx) ... (initInterTab2D's 15-bit table: exact integers for the bilinear
    kernel, their sum is 2^15, the table's correction step never fires); uint16: float32 products with the float table (1 - fy/32)(1 -
    fx/32) ..., summed left to right in float32, cvRound, saturate."""
    Minv = cv_invert_affine(M_fwd)
    ys, xs = np.meshgrid(np.arange(out_h) + y0, np.arange(out_w) + x0, indexing='ij')
    sx, sy, fx, fy = cv_fixed_coords(Minv, xs, ys)
    h, w = src.shape

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return np.where(ok, src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], 0)
    v = [tap(sy, sx), tap(sy, sx + 1), tap(sy + 1, sx), tap(sy + 1, sx + 1)]
    if src.dtype == np.uint8:
        wt = [32 * (32 - fy) * (32 - fx), 32 * (32 - fy) * fx, 32 * fy * (32 - fx), 32 * fy * fx]
        acc = sum(a.astype(np.int64) * b for a, b in zip(v, wt))
        return np.clip((acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS, 0, 255).astype(np.uint8)
    one, sc = np.float32(1.0), np.float32(1.0 / INTER_TAB_SIZE)
    cy = [one - fy.astype(np.float32) * sc, fy.astype(np.float32) * sc]
    cx = [one - fx.astype(np.float32) * sc, fx.astype(np.float32) * sc]
    wt = [cy[0] * cx[0], cy[0] * cx[1], cy[1] * cx[0], cy[1] * cx[1]]  # float32 (exact: multiples of 1/1024)
    acc = v[0].astype(np.float32) * wt[0]
    for a, b in zip(v[1:], wt[1:]):
        acc = acc + a.astype(np.float32) * b  # float32 multiply, float32 add, in this order
    return np.clip(np.rint(acc), 0, 65535).astype(np.uint16)


def rotate_clean_3D_xy(vol, angle_deg):
    """__rotate_clean_3D_xy (base_dataset.py:447-453): every z-slice rotated (cv2.warpAffine, restated) and cropped to the inscribed
    rectangle."""
    plan = clean_rotation_plan(vol.shape[1], vol.shape[2], angle_deg)
    x1, y1, x2, y2 = plan['rect']
    return np.stack([warp_affine_cv(s, plan['affine'], x1, y1, y2 - y1, x2 - x1) for s in vol])
