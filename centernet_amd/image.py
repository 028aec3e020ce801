"""Host-side geometry helpers (mirror of src/lib/utils/image.py:19-66).

OpenCV is not available in this image, so the three cv2 calls the reference makes are restated
from OpenCV's published algorithms (the restatement with citations lives in oracle/pre_oracle.py;
everything here is tested bit for bit against it):
``cv2.getAffineTransform`` -- the reference's three float32 point pairs, solved in float64 by LU
with partial pivoting (the 6 x 6 system of cv::getAffineTransform splits into two 3 x 3 systems
with the same matrix, eliminated identically); ``cv2.warpAffine(..., INTER_LINEAR)`` and
``cv2.resize`` -- OpenCV's uint8 fixed-point bilinear (1/32-pixel sample positions, 15-bit /
11-bit weights), in the library's host routines ``cn_warp_affine_u8_host`` /
``cn_resize_linear_u8_host`` and, for frames that are already on the device, in csrc/cn_pre.hip.
"""
import numpy as np


def _lu3(a, r):
    """Solve the 3 x 3 system a @ x = r: Gaussian elimination with partial pivoting (the first
    row holding the largest |pivot| wins), separate multiply / add, back substitution left to
    right -- the row operations cv::solve(DECOMP_LU) performs (core/src/matrix_decomp.cpp)."""
    a = [[float(v) for v in row] for row in a]
    r = [float(v) for v in r]
    for i in range(3):
        k = max(range(i, 3), key=lambda j: (abs(a[j][i]), -j))
        if abs(a[k][i]) < 2.220446049250313e-14:
            raise ZeroDivisionError("degenerate point triple")
        if k != i:
            a[i], a[k] = a[k], a[i]
            r[i], r[k] = r[k], r[i]
        d = -1.0 / a[i][i]
        for j in range(i + 1, 3):
            alpha = a[j][i] * d
            for c in range(i + 1, 3):
                a[j][c] = a[j][c] + alpha * a[i][c]
            r[j] = r[j] + alpha * r[i]
    x = [0.0, 0.0, 0.0]
    for i in (2, 1, 0):
        s = r[i]
        for c in range(i + 1, 3):
            s = s - a[i][c] * x[c]
        x[i] = s / a[i][i]
    return x


def affine_through(src, dst):
    """cv2.getAffineTransform(np.float32(src), np.float32(dst)): the 2 x 3 float64 matrix that
    takes three source points onto three destination points.

    OpenCV eliminates ONE 6 x 6 system whose rows alternate (x-equation, y-equation) of points
    0, 1, 2.  The two coordinate blocks never mix (each row is zero in the other block's
    columns), so the result is two 3 x 3 eliminations of the same matrix [x y 1] -- but not in
    the same row order: while columns 0-2 are eliminated, the pivot search of column 1 always
    finds row 1 (the y-equation of point 0, zero there) in its way and swaps it down, and the
    y-equations come out of that phase ordered (point 1, point 0, point 2).  Pivot ties -- the
    reference's points 0 and 1 share their x for rot = 0 -- are broken by row order, so the order
    is part of the result, bit for bit."""
    s = np.asarray(src, np.float32).astype(np.float64)
    d = np.asarray(dst, np.float32).astype(np.float64)
    row = [[s[i, 0], s[i, 1], 1.0] for i in range(3)]
    mx = _lu3([row[0], row[1], row[2]], [d[0, 0], d[1, 0], d[2, 0]])
    my = _lu3([row[1], row[0], row[2]], [d[1, 1], d[0, 1], d[2, 1]])
    return np.array([mx, my], np.float64)


def get_affine_transform(center, scale, rot, output_size, shift=(0.0, 0.0), inv=0):
    """2x3 float64 matrix between the source frame and an ``output_size`` = (w, h) crop
    (call-compatible with utils/image.py:27-60).

    The reference pins the map with three point pairs held in float32 arrays -- the crop centre,
    a point half the source x-extent above it (turned by ``rot``), and a third at a right angle
    to those two -- and asks OpenCV for the affine through them.  The float32 rounding of the
    points is part of the result (it moves the matrix by ~1e-5 px at the image border, enough to
    change which 1/32-pixel position a warp samples), so the same construction is used here."""
    extent = np.asarray(scale, np.float32).reshape(-1)
    if extent.size == 1:
        extent = np.repeat(extent, 2)
    theta = np.pi * rot / 180
    sn, cs = np.sin(theta), np.cos(theta)
    up = extent[0] * -0.5                               # float32, as the reference's src_w * -0.5
    offset = extent * np.asarray(shift, np.float32)
    pts = np.zeros((2, 3, 2), np.float32)               # [source | crop][point][x, y]
    pts[0, 0] = np.asarray(center, np.float32) + offset
    pts[0, 1] = np.asarray(center, np.float32) + np.array([-up * sn, up * cs]) + offset
    pts[1, 0] = [output_size[0] * 0.5, output_size[1] * 0.5]
    pts[1, 1] = pts[1, 0] + np.array([0, output_size[0] * -0.5], np.float32)
    for side in pts:                                    # third point: second + perpendicular
        delta = side[0] - side[1]
        side[2] = side[1] + np.array([-delta[1], delta[0]], np.float32)
    return affine_through(pts[1], pts[0]) if inv else affine_through(pts[0], pts[1])


def apply_affine(points, trans):
    """(N, 2) points through a 2x3 matrix: float32 homogeneous coordinates, float64 product
    (the arithmetic of utils/image.py:63-66 for a whole array at once)."""
    p = np.asarray(points, np.float32).reshape(-1, 2).astype(np.float64)
    t = np.asarray(trans, np.float64)
    # the dot product of utils/image.py:65 written out term by term ((t0*x + t1*y) + t2*1): plain
    # element-wise numpy, no BLAS call (and none of its worker threads) on the per-frame path
    out = np.empty((p.shape[0], 2), np.float64)
    out[:, 0] = p[:, 0] * t[0, 0] + p[:, 1] * t[0, 1] + t[0, 2]
    out[:, 1] = p[:, 0] * t[1, 0] + p[:, 1] * t[1, 1] + t[1, 2]
    return out


def affine_transform(pt, t):
    """One point (utils/image.py:63-66)."""
    return apply_affine(np.asarray(pt, np.float32).reshape(1, 2), t)[0]


def transform_preds(coords, center, scale, output_size):
    """Output-grid coordinates -> source-frame coordinates (utils/image.py:19-24): one inverse
    map for the whole (K, 2) array instead of the reference's per-point loop."""
    to_source = get_affine_transform(center, scale, 0, output_size, inv=1)
    target = np.zeros(coords.shape)
    target[:, 0:2] = apply_affine(coords[:, 0:2], to_source)
    return target


def invert_affine(trans):
    """dst -> src 2x3 float64 matrix of a src -> dst 2x3 affine, formed the way cv::warpAffine
    forms it from its M (closed 2 x 2 inverse, then the translation from the inverted entries)."""
    m00, m01, m02, m10, m11, m12 = (float(v) for v in np.asarray(trans, np.float64).reshape(-1)[:6])
    det = m00 * m11 - m01 * m10
    r = 1.0 / det if det != 0.0 else 0.0
    i00, i01, i10, i11 = m11 * r, m01 * -r, m10 * -r, m00 * r
    return np.array([[i00, i01, -i00 * m02 - i01 * m12],
                     [i10, i11, -i10 * m02 - i11 * m12]], np.float64)


def _host_u8(img):
    img = np.ascontiguousarray(img)
    if img.dtype != np.uint8 or img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] > 4):
        raise ValueError("uint8 (H, W) or (H, W, C <= 4) images only")
    return img, (1 if img.ndim == 2 else img.shape[2])


def warp_affine(img, trans, dsize):
    """cv2.warpAffine(img, trans, dsize, flags=cv2.INTER_LINEAR) for uint8 images, zero border
    (library host routine ``cn_warp_affine_u8_host``)."""
    import ctypes
    from . import native
    img, ch = _host_u8(img)
    w_out, h_out = int(dsize[0]), int(dsize[1])
    out = np.empty((h_out, w_out) + img.shape[2:], np.uint8)
    m = (ctypes.c_double * 6)(*invert_affine(trans).reshape(-1))
    native.check(native.lib().cn_warp_affine_u8_host(
        img.ctypes.data_as(ctypes.c_void_p), img.shape[0], img.shape[1], ch, m, h_out, w_out,
        out.ctypes.data_as(ctypes.c_void_p)), "cn_warp_affine_u8_host")
    return out


def resize_bilinear(img, dsize):
    """cv2.resize(img, (w, h)) with the default INTER_LINEAR for uint8 images (library host
    routine ``cn_resize_linear_u8_host``)."""
    import ctypes
    from . import native
    img, ch = _host_u8(img)
    w_out, h_out = int(dsize[0]), int(dsize[1])
    out = np.empty((h_out, w_out) + img.shape[2:], np.uint8)
    native.check(native.lib().cn_resize_linear_u8_host(
        img.ctypes.data_as(ctypes.c_void_p), img.shape[0], img.shape[1], ch, h_out, w_out,
        out.ctypes.data_as(ctypes.c_void_p)), "cn_resize_linear_u8_host")
    return out


def normalize_chw(inp_u8, mean, std):
    """((inp / 255. - mean) / std).astype(float32) then HWC -> CHW (base_detector.py:56-58), in the
    library's host routine (numpy's float64 arithmetic through a 256-entry table per channel;
    ``normalize_chw_numpy`` is the definition it is checked against)."""
    import ctypes
    from . import native
    img = np.ascontiguousarray(inp_u8)
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
        return normalize_chw_numpy(inp_u8, mean, std)
    m = np.ascontiguousarray(np.asarray(mean, np.float32).reshape(3))
    sd = np.ascontiguousarray(np.asarray(std, np.float32).reshape(3))
    out = np.empty((3, img.shape[0], img.shape[1]), np.float32)
    native.check(native.lib().cn_normalize_u8_chw_f32_host(
        img.ctypes.data_as(ctypes.c_void_p), img.shape[0], img.shape[1],
        m.ctypes.data_as(ctypes.c_void_p), sd.ctypes.data_as(ctypes.c_void_p),
        out.ctypes.data_as(ctypes.c_void_p)), "cn_normalize_u8_chw_f32_host")
    return out


def normalize_chw_numpy(inp_u8, mean, std):
    """``normalize_chw`` as the reference writes it (base_detector.py:56-58)."""
    mean = np.asarray(mean, np.float32).reshape(1, 1, 3)
    std = np.asarray(std, np.float32).reshape(1, 1, 3)
    inp = ((inp_u8 / 255. - mean) / std).astype(np.float32)
    return inp.transpose(2, 0, 1)


def flip(img):
    return img[:, :, ::-1].copy()
