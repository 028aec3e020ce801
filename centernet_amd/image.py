"""Host-side geometry helpers (mirror of src/lib/utils/image.py:19-66).

OpenCV is not available in this image, so the cv2 calls the reference makes are restated:
``cv2.getAffineTransform`` (exact 3-point solve, float64), ``cv2.warpAffine(..., INTER_LINEAR)``
and ``cv2.resize`` (float64 bilinear, round-half-even to uint8; the same arithmetic, operation for
operation, as the device kernels in csrc/cn_pre.hip).  For the benchmark configuration
(512x512 input, fix_res) the warp is the identity.
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


def invert_affine(trans):
    """dst -> src 2x3 float64 matrix of a src -> dst 2x3 affine (what cv2.warpAffine does
    with its M unless WARP_INVERSE_MAP is set)."""
    M = np.vstack([np.asarray(trans, np.float64), [0, 0, 1]])
    return np.linalg.inv(M)[:2].copy()


def resize_matrix(in_size, out_size):
    """dst -> src matrix of cv2.resize(INTER_LINEAR): src = (dst + 0.5) * (in/out) - 0.5."""
    (w_in, h_in), (w_out, h_out) = in_size, out_size
    sx, sy = float(w_in) / float(w_out), float(h_in) / float(h_out)
    return np.array([[sx, 0.0, 0.5 * sx - 0.5], [0.0, sy, 0.5 * sy - 0.5]], np.float64)


def warp_bilinear_u8(img, Mi, dsize, replicate=False):
    """The arithmetic contract shared with the device kernel (csrc/cn_pre.hip) and
    oracle/pre_oracle.py: float64 bilinear in a fixed operation order, taps outside the image
    zero (or clamped when ``replicate``), round-half-even to uint8."""
    w_out, h_out = int(dsize[0]), int(dsize[1])
    h_in, w_in = img.shape[:2]
    Mi = np.asarray(Mi, np.float64)
    xs, ys = np.meshgrid(np.arange(w_out, dtype=np.float64), np.arange(h_out, dtype=np.float64))
    sx = (Mi[0, 0] * xs + Mi[0, 1] * ys) + Mi[0, 2]
    sy = (Mi[1, 0] * xs + Mi[1, 1] * ys) + Mi[1, 2]
    fx0, fy0 = np.floor(sx), np.floor(sy)
    far = ~((fx0 > -4.0) & (fx0 < w_in + 4.0) & (fy0 > -4.0) & (fy0 < h_in + 4.0))
    x0 = np.where(far, -4, fx0).astype(np.int64)
    y0 = np.where(far, -4, fy0).astype(np.int64)
    fx = np.where(far, 0.0, sx - fx0)[..., None]
    fy = np.where(far, 0.0, sy - fy0)[..., None]
    gx, gy = 1.0 - fx, 1.0 - fy
    src = img.astype(np.float64)
    if src.ndim == 2:
        src = src[..., None]

    def tap(yy, xx):
        if replicate:
            ok = ~far
        else:
            ok = (yy >= 0) & (yy < h_in) & (xx >= 0) & (xx < w_in)
        v = src[np.clip(yy, 0, h_in - 1), np.clip(xx, 0, w_in - 1)]
        return v * ok[..., None]

    out = (tap(y0, x0) * gx) * gy
    out = out + (tap(y0, x0 + 1) * fx) * gy
    out = out + (tap(y0 + 1, x0) * gx) * fy
    out = out + (tap(y0 + 1, x0 + 1) * fx) * fy
    out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[..., 0]


def warp_affine(img, trans, dsize):
    """cv2.warpAffine(img, trans, dsize, flags=INTER_LINEAR), zero border (float bilinear; see
    warp_bilinear_u8 for the exact arithmetic).  uint8 images only."""
    assert img.dtype == np.uint8, "warp_affine restates the uint8 path the detectors use"
    return warp_bilinear_u8(img, invert_affine(trans), dsize, replicate=False)


def resize_bilinear(img, dsize):
    """cv2.resize(img, (w, h)) with INTER_LINEAR (half-pixel centres, replicated border)."""
    w_out, h_out = int(dsize[0]), int(dsize[1])
    h_in, w_in = img.shape[:2]
    if (h_in, w_in) == (h_out, w_out):
        return img.copy()
    assert img.dtype == np.uint8
    return warp_bilinear_u8(img, resize_matrix((w_in, h_in), (w_out, h_out)), (w_out, h_out),
                            replicate=True)


def normalize_chw(inp_u8, mean, std):
    """((inp / 255. - mean) / std).astype(float32) then HWC -> CHW (base_detector.py:56-58)."""
    mean = np.asarray(mean, np.float32).reshape(1, 1, 3)
    std = np.asarray(std, np.float32).reshape(1, 1, 3)
    inp = ((inp_u8 / 255. - mean) / std).astype(np.float32)
    return inp.transpose(2, 0, 1)


def flip(img):
    return img[:, :, ::-1].copy()
