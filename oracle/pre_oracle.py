"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of BaseDetector.pre_process (src/lib/detectors/base_detector.py:37-65):

    trans_input = get_affine_transform(c, s, 0, [inp_w, inp_h])      utils/image.py:27-60
    resized     = cv2.resize(image, (new_w, new_h))                  base_detector.py:52
    inp         = cv2.warpAffine(resized, trans_input, (inp_w, inp_h), flags=cv2.INTER_LINEAR)
    inp         = ((inp / 255. - mean) / std).astype(np.float32)     :56

The arithmetic of the three cv2 calls lives in a third-party dependency that is absent from
/root/reference and from this image (requirements.txt:1 `opencv-python`, unpinned).  SURVEY 8(c)
/ the task rules: restate its PUBLISHED algorithm.  What is restated here is OpenCV's uint8
INTER_LINEAR path as published in modules/imgproc/src/imgwarp.cpp and resize.cpp (3.4 / 4.x;
the portable C++ loops; the SIMD branches next to them are written to produce the same
integers; a build that routes these calls to IPP or OpenCL may differ in the last bit):

  cv::getAffineTransform   6x6 system through three point pairs, solve() = LU with partial
                           pivoting in double (core/src/matrix_decomp.cpp LUImpl)
  cv::warpAffine           M inverted in double by the closed 2x2 formula; per destination pixel
                           X = (round((M1*y + M2)*1024) + 16 + round(M0*x*1024)) >> 5 (same for Y):
                           source position in 1/32 pixel (AB_BITS = 10, INTER_BITS = 5); integer
                           part = top-left tap, 5-bit fractions index a 32 x 32 table of four int16
                           weights (INTER_REMAP_COEF_BITS = 15, sum 32768; the entry for fraction
                           (0,0) is [32767, 0, 0, 1]: 1.0 does not fit int16 and the fix-up loop
                           of initInterTab2D repairs the sum on tap 3); taps outside the image are
                           the border value 0; dst = sat_u8((sum + 2^14) >> 15)
  cv::resize INTER_LINEAR  same size: copy.  Exactly half size in both directions: the 2x2 area
                           mean (a + b + c + d + 2) >> 2 (resize.cpp switches to INTER_AREA).
                           Otherwise separable: fx = float((dx + 0.5)*scale - 0.5), left tap
                           floor(fx), coefficients saturate_cast<short>((1-f)*2048) and (f*2048)
                           (INTER_RESIZE_COEF_BITS = 11, each rounded on its own), columns clamped
                           with f = 0, rows clamped without touching f; horizontal pass exact
                           integers, vertical pass the uint8 specialisation
                           (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.

PARITY: pinned by hand-computed fixed-point cases (tests/test_oracle_pre.py: each expected value
is derived in the test from the constants above with integer arithmetic a reader can redo on
paper), NOT by outputs of an OpenCV build -- none is available offline.  The restatement is
mine, from the published sources, and is the single definition the device kernels
(csrc/cn_pre.hip) and the host routines (cn_warp_affine_u8_host, cn_resize_linear_u8_host) are
held to, bit for bit.
"""
import numpy as np

AB_BITS = 10
INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
INTER_REMAP_COEF_BITS = 15
INTER_RESIZE_COEF_BITS = 11


def cv_round(v):
    """cvRound(double): round to nearest, ties to even (cvtsd2si / lrint), vectorised."""
    return np.rint(np.asarray(v, np.float64)).astype(np.int64)


def _sat_short(v):
    return np.clip(v, -32768, 32767)


# ---------------------------------------------------------------------------------------------
# cv::getAffineTransform
def lu_solve(a, b):
    """core/src/matrix_decomp.cpp LUImpl<double>: in-place elimination with partial pivoting,
    then back substitution; separate multiplies and adds (no FMA), left-to-right sums."""
    a = np.array(a, np.float64)
    b = np.array(b, np.float64).reshape(-1)
    m = a.shape[0]
    for i in range(m):
        k = i
        for j in range(i + 1, m):
            if abs(a[j, i]) > abs(a[k, i]):
                k = j
        if abs(a[k, i]) < np.finfo(np.float64).eps * 100:
            raise ZeroDivisionError("singular system")
        if k != i:
            a[[i, k], i:] = a[[k, i], i:]
            b[[i, k]] = b[[k, i]]
        d = -1.0 / a[i, i]
        for j in range(i + 1, m):
            alpha = a[j, i] * d
            for kk in range(i + 1, m):
                a[j, kk] = a[j, kk] + alpha * a[i, kk]
            b[j] = b[j] + alpha * b[i]
    for i in range(m - 1, -1, -1):
        s = b[i]
        for k in range(i + 1, m):
            s = s - a[i, k] * b[k]
        b[i] = s / a[i, i]
    return b


def cv_get_affine_transform(src, dst):
    """cv::getAffineTransform(const Point2f src[3], const Point2f dst[3]) -> 2x3 float64."""
    src = np.asarray(src, np.float32).astype(np.float64)
    dst = np.asarray(dst, np.float32).astype(np.float64)
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for i in range(3):
        a[2 * i, 0:3] = [src[i, 0], src[i, 1], 1.0]
        a[2 * i + 1, 3:6] = [src[i, 0], src[i, 1], 1.0]
        b[2 * i], b[2 * i + 1] = dst[i, 0], dst[i, 1]
    return lu_solve(a, b).reshape(2, 3)


def get_affine_transform(center, scale, rot, output_size, shift=(0.0, 0.0), inv=0):
    """utils/image.py:27-60: the three point pairs in float32 arrays, then cv2.getAffineTransform."""
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale], dtype=np.float32)
    scale_tmp = np.asarray(scale)
    shift = np.asarray(shift, np.float32)
    src_w, dst_w, dst_h = scale_tmp[0], output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    p = [0, src_w * -0.5]
    src_dir = [p[0] * cs - p[1] * sn, p[0] * sn + p[1] * cs]      # get_dir, :69-77
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir

    def third(a, b):                                               # get_3rd_point, :80-82
        direct = a - b
        return b + np.array([-direct[1], direct[0]], dtype=np.float32)
    src[2:, :] = third(src[0, :], src[1, :])
    dst[2:, :] = third(dst[0, :], dst[1, :])
    return cv_get_affine_transform(dst, src) if inv else cv_get_affine_transform(src, dst)


# ---------------------------------------------------------------------------------------------
# cv::warpAffine, uint8, INTER_LINEAR, BORDER_CONSTANT(0)
def cv_invert_affine(M):
    """The in-place inversion at the top of cv::warpAffine (imgwarp.cpp, no WARP_INVERSE_MAP)."""
    m = [float(v) for v in np.asarray(M, np.float64).reshape(-1)[:6]]
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return np.array(m, np.float64).reshape(2, 3)


def cv_bilinear_tab():
    """initInterTab2D(INTER_LINEAR, fixpt = true): (1024, 4) int weights [tl, tr, bl, br] for the
    fraction index fy * 32 + fx."""
    tab = np.zeros((INTER_TAB_SIZE * INTER_TAB_SIZE, 4), np.int64)
    one = np.float32(1.0)
    sc = np.float32(1.0) / np.float32(INTER_TAB_SIZE)
    t1 = [(one - np.float32(i) * sc, np.float32(i) * sc) for i in range(INTER_TAB_SIZE)]
    for i in range(INTER_TAB_SIZE):          # y fraction
        for j in range(INTER_TAB_SIZE):      # x fraction
            it = []
            for k1 in range(2):
                for k2 in range(2):
                    v = np.float32(t1[i][k1] * t1[j][k2])
                    it.append(int(_sat_short(cv_round(np.float64(v) * 32768.0))))
            isum = sum(it)
            if isum != 32768:
                # fix-up loop of initInterTab2D with ksize = 2: ksize2 = 1, it scans k1, k2 in
                # {1, 2} -- for a 2x2 kernel those are tap 3 and the first entries of the NEXT
                # table cell, still zero at that point -- so the largest / smallest candidate is
                # tap 3 and the whole difference lands there
                diff = isum - 32768
                it[3] -= diff
            tab[i * INTER_TAB_SIZE + j] = it
    return tab


_TAB = None


def cv_warp_affine_u8(img, M, dsize, inverse_map=False):
    """cv2.warpAffine(img, M, dsize, flags=cv2.INTER_LINEAR [| cv2.WARP_INVERSE_MAP]) for uint8
    (H, W, C) / (H, W)."""
    global _TAB
    if _TAB is None:
        _TAB = cv_bilinear_tab()
    img = np.asarray(img)
    assert img.dtype == np.uint8
    w_out, h_out = int(dsize[0]), int(dsize[1])
    h_in, w_in = img.shape[:2]
    src = img.reshape(h_in, w_in, -1).astype(np.int64)
    m = (np.asarray(M, np.float64) if inverse_map else cv_invert_affine(M)).reshape(-1)
    AB_SCALE = 1 << AB_BITS
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    xs = np.arange(w_out, dtype=np.float64)
    ys = np.arange(h_out, dtype=np.float64)
    adelta = cv_round((m[0] * xs) * AB_SCALE)
    bdelta = cv_round((m[3] * xs) * AB_SCALE)
    X0 = cv_round((m[1] * ys + m[2]) * AB_SCALE) + round_delta
    Y0 = cv_round((m[4] * ys + m[5]) * AB_SCALE) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)     # arithmetic shifts
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = _sat_short(X >> INTER_BITS)
    sy = _sat_short(Y >> INTER_BITS)
    w = _TAB[(Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1))]   # (h, w, 4)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h_in) & (xx >= 0) & (xx < w_in)
        v = src[np.clip(yy, 0, h_in - 1), np.clip(xx, 0, w_in - 1)]
        return v * ok[..., None]
    acc = tap(sy, sx) * w[..., 0:1] + tap(sy, sx + 1) * w[..., 1:2] + \
        tap(sy + 1, sx) * w[..., 2:3] + tap(sy + 1, sx + 1) * w[..., 3:4]
    out = np.clip((acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS, 0, 255)
    out = out.astype(np.uint8)
    return out.reshape((h_out, w_out) + img.shape[2:])


# ---------------------------------------------------------------------------------------------
# cv::resize, uint8, INTER_LINEAR
def _resize_axis(n_in, n_out, clamp_fraction):
    """Left tap index and the two int16 coefficients per destination index (resize.cpp, the
    xofs / ialpha and yofs / ibeta loops)."""
    scale = 1.0 / (float(n_out) / float(n_in))            # scale_x = 1. / inv_scale_x
    d = np.arange(n_out, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_fraction:                                    # columns only
        lo = s < 0
        f = np.where(lo, np.float32(0), f)
        s = np.where(lo, 0, s)
        hi = s >= n_in - 1
        f = np.where(hi, np.float32(0), f)
        s = np.where(hi, n_in - 1, s)
    c0 = _sat_short(cv_round((np.float32(1.0) - f).astype(np.float64) * 2048.0))
    c1 = _sat_short(cv_round(f.astype(np.float64) * 2048.0))
    return s, c0, c1


def cv_resize_linear_u8(img, dsize):
    """cv2.resize(img, (w, h)) with the default INTER_LINEAR, uint8 (H, W, C) / (H, W)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    w_out, h_out = int(dsize[0]), int(dsize[1])
    h_in, w_in = img.shape[:2]
    if (h_in, w_in) == (h_out, w_out):
        return img.copy()
    src = img.reshape(h_in, w_in, -1).astype(np.int64)
    if h_in == 2 * h_out and w_in == 2 * w_out:          # INTER_LINEAR -> INTER_AREA fast path
        s = src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2]
        return ((s + 2) >> 2).astype(np.uint8).reshape((h_out, w_out) + img.shape[2:])
    sx, a0, a1 = _resize_axis(w_in, w_out, True)
    sy, b0, b1 = _resize_axis(h_in, h_out, False)
    sx1 = np.minimum(sx + 1, w_in - 1)                   # a1 = 0 wherever sx + 1 is outside
    rows = src[:, sx] * a0[None, :, None] + src[:, sx1] * a1[None, :, None]      # (h_in, w_out, C)
    r0 = np.clip(sy, 0, h_in - 1)
    r1 = np.clip(sy + 1, 0, h_in - 1)
    S0, S1 = rows[r0], rows[r1]
    out = (((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8).reshape((h_out, w_out) + img.shape[2:])


# ---------------------------------------------------------------------------------------------
def pre_process(image, scale, mean, std, fix_res=True, input_h=512, input_w=512, pad=31,
                flip_test=False, down_ratio=4):
    """base_detector.py:37-65 -> (images (1|2,3,H,W) float32, meta)."""
    height, width = image.shape[0:2]
    new_height, new_width = int(height * scale), int(width * scale)
    if fix_res:
        inp_height, inp_width = input_h, input_w
        c = np.array([new_width / 2., new_height / 2.], dtype=np.float32)
        s = max(height, width) * 1.0
    else:
        inp_height, inp_width = (new_height | pad) + 1, (new_width | pad) + 1
        c = np.array([new_width // 2, new_height // 2], dtype=np.float32)
        s = np.array([inp_width, inp_height], dtype=np.float32)
    trans = get_affine_transform(c, s, 0, [inp_width, inp_height])
    resized = cv_resize_linear_u8(image, (new_width, new_height))
    inp = cv_warp_affine_u8(resized, trans, (inp_width, inp_height))
    mean = np.asarray(mean, np.float32).reshape(1, 1, 3)
    std = np.asarray(std, np.float32).reshape(1, 1, 3)
    inp = ((inp / 255. - mean) / std).astype(np.float32)
    images = inp.transpose(2, 0, 1).reshape(1, 3, inp_height, inp_width)
    if flip_test:
        images = np.concatenate((images, images[:, :, :, ::-1]), axis=0)
    meta = {'c': c, 's': s, 'out_height': inp_height // down_ratio, 'out_width': inp_width // down_ratio}
    return np.ascontiguousarray(images), meta


def ddd_pre_process(image, mean, std, input_h=384, input_w=1280, keep_res=False, down_ratio=4, calib=None):
    """DddDetector.pre_process, detectors/ddd.py:30-54: no resize, the frame warped straight onto the
    fixed input, FLOAT32 normalisation ``(x.astype(float32) / 255. - mean) / std`` -> (images
    (1, 3, H, W) float32, meta without the detector's default calib filled in)."""
    height, width = image.shape[0:2]
    c = np.array([width / 2, height / 2], dtype=np.float32)
    s = np.array([input_w, input_h], dtype=np.int32) if keep_res else np.array([width, height], dtype=np.int32)
    trans = get_affine_transform(c, s, 0, [input_w, input_h])
    inp = cv_warp_affine_u8(image, trans, (input_w, input_h))
    inp = inp.astype(np.float32) / 255.
    inp = (inp - np.asarray(mean, np.float32).reshape(1, 1, 3)) / np.asarray(std, np.float32).reshape(1, 1, 3)
    images = inp.transpose(2, 0, 1)[np.newaxis, ...]
    meta = {'c': c, 's': s, 'out_height': input_h // down_ratio, 'out_width': input_w // down_ratio,
            'calib': None if calib is None else np.array(calib, dtype=np.float32)}
    return np.ascontiguousarray(images), meta
