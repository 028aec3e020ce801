"""Host-side geometry helpers (mirror of src/lib/utils/image.py:19-66).

OpenCV is not available in this image, so the cv2 calls the reference makes are restated:
``cv2.getAffineTransform`` of the reference's three-point construction (a similarity, written
in closed form in float64), ``cv2.warpAffine(..., INTER_LINEAR)``
and ``cv2.resize`` (float64 bilinear, round-half-even to uint8; the same arithmetic, operation for
operation, as the device kernels in csrc/cn_pre.hip).  For the benchmark configuration
(512x512 input, fix_res) the warp is the identity.
"""
import numpy as np


def _rotation(rot_deg):
    """(cos, sin) of the augmentation angle; exactly (1, 0) for the inference case rot = 0."""
    theta = np.pi * rot_deg / 180
    return float(np.cos(theta)), float(np.sin(theta))


def get_affine_transform(center, scale, rot, output_size, shift=(0.0, 0.0), inv=0):
    """2x3 float64 matrix between the source frame and an ``output_size`` = (w, h) crop
    (call-compatible with utils/image.py:27-60).

    The reference pins the map with three point pairs -- the crop centre, a point half the
    source extent above it (turned by ``rot``), and a third at a right angle -- and asks OpenCV
    for the affine through them.  Right-angle constructions on both sides make that map a
    similarity, so it is written down directly here:

        dst = q0 + k * R(-rot) * (src - p0),      k = dst_w / src_w

    with p0 = centre + scale * shift (held in float32, as the reference's point array holds
    it), q0 = the crop centre, and src_w = scale[0] (the reference uses the x extent for both
    axes).  ``inv`` returns the opposite direction, src = p0 + R(rot) * (dst - q0) / k.  For
    rot = 0 -- every call on the inference path -- the entries are exact ratios of the inputs.
    """
    extent = np.asarray(scale, np.float32).reshape(-1)
    if extent.size == 1:
        extent = np.repeat(extent, 2)
    p0 = (np.asarray(center, np.float32).reshape(2) +
          extent * np.asarray(shift, np.float32).reshape(2)).astype(np.float64)
    q0 = np.array([output_size[0] * 0.5, output_size[1] * 0.5], np.float64)
    cs, sn = _rotation(rot)
    if inv:
        k = float(extent[0]) / float(output_size[0])
        lin = k * np.array([[cs, -sn], [sn, cs]], np.float64)
        return np.concatenate([lin, (p0 - lin @ q0)[:, None]], axis=1)
    k = float(output_size[0]) / float(extent[0])
    lin = k * np.array([[cs, sn], [-sn, cs]], np.float64)
    return np.concatenate([lin, (q0 - lin @ p0)[:, None]], axis=1)


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
    zero (or clamped when ``replicate``), round-half-even to uint8.  Runs in the library's host
    routine ``cn_warp_bilinear_u8_host`` (the same operations in C, ~2 ms per 512x512 frame);
    ``warp_bilinear_u8_numpy`` below is the array-at-once statement of it (30 ms), kept as the
    readable definition and checked against it bit for bit (tests/test_host.py)."""
    import ctypes
    from . import native
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim in (2, 3)
    w_out, h_out = int(dsize[0]), int(dsize[1])
    ch = 1 if img.ndim == 2 else img.shape[2]
    if ch > 4:
        return warp_bilinear_u8_numpy(img, Mi, dsize, replicate)
    out = np.empty((h_out, w_out) if img.ndim == 2 else (h_out, w_out, ch), np.uint8)
    m = (ctypes.c_double * 6)(*np.asarray(Mi, np.float64).reshape(-1)[:6])
    native.check(native.lib().cn_warp_bilinear_u8_host(
        img.ctypes.data_as(ctypes.c_void_p), img.shape[0], img.shape[1], ch, m, h_out, w_out,
        int(bool(replicate)), out.ctypes.data_as(ctypes.c_void_p)), "cn_warp_bilinear_u8_host")
    return out


def warp_bilinear_u8_numpy(img, Mi, dsize, replicate=False):
    """``warp_bilinear_u8`` written with whole-array numpy operations (the definition the C
    routine and the device kernels follow operation for operation)."""
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
