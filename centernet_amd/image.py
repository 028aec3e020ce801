"""Host-side geometry helpers (mirror of src/lib/utils/image.py:19-66).

OpenCV is not available in this image, so the two cv2 calls the reference makes are
restated: ``cv2.getAffineTransform`` (exact 3-point solve, float64) and
``cv2.warpAffine(..., INTER_LINEAR)`` (float bilinear, zero border).  For the benchmark
configuration (512x512 input, fix_res) the warp is the identity.
"""
import numpy as np


def get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]


def get_3rd_point(a, b):
    direct = a - b
    return b + np.array([-direct[1], direct[0]], dtype=np.float32)


def solve_affine(src, dst):
    """2x3 float64 matrix M with M @ [x, y, 1] = dst for the three point pairs
    (what cv2.getAffineTransform returns)."""
    src = np.asarray(src, dtype=np.float64)
    dst = np.asarray(dst, dtype=np.float64)
    A = np.concatenate([src, np.ones((3, 1))], axis=1)  # 3x3
    return np.linalg.solve(A, dst).T.copy()            # 2x3


def get_affine_transform(center, scale, rot, output_size,
                         shift=np.array([0, 0], dtype=np.float32), inv=0):
    # utils/image.py:27-60
    if not isinstance(scale, np.ndarray) and not isinstance(scale, list):
        scale = np.array([scale, scale], dtype=np.float32)
    scale_tmp = scale
    src_w = scale_tmp[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    src_dir = get_dir([0, src_w * -0.5], rot_rad)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    src[2:, :] = get_3rd_point(src[0, :], src[1, :])
    dst[2:, :] = get_3rd_point(dst[0, :], dst[1, :])
    if inv:
        return solve_affine(dst, src)
    return solve_affine(src, dst)


def affine_transform(pt, t):
    # utils/image.py:63-66
    new_pt = np.array([pt[0], pt[1], 1.], dtype=np.float32).T
    new_pt = np.dot(t, new_pt)
    return new_pt[:2]


def transform_preds(coords, center, scale, output_size):
    # utils/image.py:19-24 (vectorised: the per-point loop is a (K,3)x(3,2) product)
    trans = get_affine_transform(center, scale, 0, output_size, inv=1)
    pts = np.concatenate([np.asarray(coords[:, 0:2], dtype=np.float32),
                          np.ones((coords.shape[0], 1), np.float32)], axis=1)
    target = np.zeros(coords.shape)
    target[:, 0:2] = pts.astype(np.float64) @ trans.T
    return target


def warp_affine(img, trans, dsize):
    """Bilinear warp, zero border: dst(x,y) = src(M^-1 [x,y,1]); img HxWxC, dsize (w,h)."""
    w_out, h_out = int(dsize[0]), int(dsize[1])
    h_in, w_in = img.shape[:2]
    M = np.vstack([np.asarray(trans, np.float64), [0, 0, 1]])
    Mi = np.linalg.inv(M)
    if (abs(Mi[0, 0] - 1) < 1e-12 and abs(Mi[1, 1] - 1) < 1e-12 and abs(Mi[0, 1]) < 1e-12 and
            abs(Mi[1, 0]) < 1e-12 and abs(Mi[0, 2]) < 1e-9 and abs(Mi[1, 2]) < 1e-9 and
            (h_in, w_in) == (h_out, w_out)):
        return img.copy()
    xs, ys = np.meshgrid(np.arange(w_out, dtype=np.float64), np.arange(h_out, dtype=np.float64))
    sx = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]
    sy = Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]
    x0 = np.floor(sx).astype(np.int64)
    y0 = np.floor(sy).astype(np.int64)
    fx = (sx - x0)[..., None]
    fy = (sy - y0)[..., None]
    src = img.astype(np.float64)
    if src.ndim == 2:
        src = src[..., None]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h_in) & (xx >= 0) & (xx < w_in)
        v = src[np.clip(yy, 0, h_in - 1), np.clip(xx, 0, w_in - 1)]
        return v * ok[..., None]

    out = (tap(y0, x0) * (1 - fx) * (1 - fy) + tap(y0, x0 + 1) * fx * (1 - fy) +
           tap(y0 + 1, x0) * (1 - fx) * fy + tap(y0 + 1, x0 + 1) * fx * fy)
    if np.issubdtype(img.dtype, np.integer):
        out = np.clip(np.rint(out), np.iinfo(img.dtype).min, np.iinfo(img.dtype).max)
    out = out.astype(img.dtype)
    return out if img.ndim == 3 else out[..., 0]


def resize_bilinear(img, dsize):
    """cv2.resize(img, (w, h)) with INTER_LINEAR (half-pixel centres)."""
    w_out, h_out = int(dsize[0]), int(dsize[1])
    h_in, w_in = img.shape[:2]
    if (h_in, w_in) == (h_out, w_out):
        return img.copy()
    sx, sy = w_in / float(w_out), h_in / float(h_out)
    trans = np.array([[1.0 / sx, 0, 0.5 / sx - 0.5], [0, 1.0 / sy, 0.5 / sy - 0.5]])
    # border handling: replicate (cv2.resize clamps), so pad by edge first
    pad = np.pad(img, ((1, 1), (1, 1)) + ((0, 0),) * (img.ndim - 2), mode="edge")
    t2 = trans.copy()
    t2[0, 2] -= 1.0 / sx
    t2[1, 2] -= 1.0 / sy
    return warp_affine(pad, t2, (w_out, h_out))


def flip(img):
    return img[:, :, ::-1].copy()
