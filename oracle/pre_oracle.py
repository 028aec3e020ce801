"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of BaseDetector.pre_process (src/lib/detectors/base_detector.py:37-65) in
plain scalar Python floats (IEEE float64, no FMA), independent of the vectorised host code in
centernet_amd/image.py and of the HIP kernels in centernet_amd/csrc/cn_pre.hip.

PARITY UNPINNED against OpenCV: the reference calls cv2.resize / cv2.warpAffine / cv2.getAffineTransform
(utils/image.py:56-58, base_detector.py:51-55); OpenCV is not installed here and the reference
holds no golden image for this step.  What is pinned: (i) the identity configuration (512x512,
fix_res) is an exact copy, as in OpenCV; (ii) integer translations copy pixels exactly;
(iii) half-pixel shifts give the hand-computed 2- and 4-tap averages (tests/test_oracle_pre.py).
OpenCV's fixed-point path (coordinates in 1/32 px, 15-bit weights) can differ from float
bilinear by one uint8 level elsewhere.
"""
import math

import numpy as np


def _round_half_even_u8(v):
    r = math.floor(v)
    d = v - r
    if d > 0.5 or (d == 0.5 and (int(r) & 1)):
        r += 1
    return int(min(max(r, 0), 255))


def warp_bilinear_u8(img, Mi, dsize, replicate=False):
    """dst(x,y) = bilinear(src, Mi @ [x,y,1]); zero border, or clamped taps when replicate."""
    w_out, h_out = int(dsize[0]), int(dsize[1])
    h_in, w_in = img.shape[:2]
    m = [float(v) for v in np.asarray(Mi, np.float64).reshape(-1)[:6]]
    out = np.zeros((h_out, w_out, 3), np.uint8)
    for y in range(h_out):
        for x in range(w_out):
            sx = (m[0] * x + m[1] * y) + m[2]
            sy = (m[3] * x + m[4] * y) + m[5]
            fx0, fy0 = math.floor(sx), math.floor(sy)
            far = not (-4.0 < fx0 < w_in + 4.0 and -4.0 < fy0 < h_in + 4.0)
            if far:
                continue  # every tap is outside: 0
            x0, y0 = int(fx0), int(fy0)
            fx, fy = sx - fx0, sy - fy0
            gx, gy = 1.0 - fx, 1.0 - fy
            for c in range(3):
                def tap(yy, xx):
                    inside = 0 <= yy < h_in and 0 <= xx < w_in
                    if not inside and not replicate:
                        return 0.0
                    return float(img[min(max(yy, 0), h_in - 1), min(max(xx, 0), w_in - 1), c])
                s = (tap(y0, x0) * gx) * gy
                s = s + (tap(y0, x0 + 1) * fx) * gy
                s = s + (tap(y0 + 1, x0) * gx) * fy
                s = s + (tap(y0 + 1, x0 + 1) * fx) * fy
                out[y, x, c] = _round_half_even_u8(s)
    return out


def resize_bilinear_u8(img, dsize):
    """cv2.resize(img, dsize) INTER_LINEAR: src = (dst + 0.5) * (in/out) - 0.5, border replicated."""
    w_out, h_out = int(dsize[0]), int(dsize[1])
    h_in, w_in = img.shape[:2]
    sx, sy = float(w_in) / float(w_out), float(h_in) / float(h_out)
    return warp_bilinear_u8(img, [sx, 0.0, 0.5 * sx - 0.5, 0.0, sy, 0.5 * sy - 0.5],
                            (w_out, h_out), replicate=True)


def solve_affine3(src, dst):
    """cv2.getAffineTransform: exact 2x3 solve through three point pairs (Cramer's rule)."""
    (x0, y0), (x1, y1), (x2, y2) = [(float(a), float(b)) for a, b in src]
    det = x0 * (y1 - y2) - y0 * (x1 - x2) + (x1 * y2 - x2 * y1)
    rows = []
    for k in range(2):
        u0, u1, u2 = float(dst[0][k]), float(dst[1][k]), float(dst[2][k])
        a = (u0 * (y1 - y2) - y0 * (u1 - u2) + (u1 * y2 - u2 * y1)) / det
        b = (x0 * (u1 - u2) - u0 * (x1 - x2) + (x1 * u2 - x2 * u1)) / det
        c = (x0 * (y1 * u2 - y2 * u1) - y0 * (x1 * u2 - x2 * u1) + u0 * (x1 * y2 - x2 * y1)) / det
        rows.append([a, b, c])
    return np.array(rows, np.float64)


def input_transform(c, s, inp_w, inp_h):
    """get_affine_transform(c, s, 0, [inp_w, inp_h]) (utils/image.py:27-60, rot = 0, shift = 0):
    source points centre, centre - (0, s_w/2), third by the 90-degree rule (:15-17)."""
    sw = float(s[0]) if isinstance(s, (np.ndarray, list, tuple)) else float(s)
    src0 = np.array([c[0], c[1]], np.float32)
    src1 = src0 + np.array([0, sw * -0.5], np.float32)
    dst0 = np.array([inp_w * 0.5, inp_h * 0.5], np.float32)
    dst1 = dst0 + np.array([0, inp_w * -0.5], np.float32)

    def third(a, b):
        d = a - b
        return b + np.array([-d[1], d[0]], np.float32)
    src = [src0, src1, third(src0, src1)]
    dst = [dst0, dst1, third(dst0, dst1)]
    return solve_affine3(src, dst)


def invert2x3(t):
    a, b, c, d, e, f = [float(v) for v in np.asarray(t).reshape(-1)]
    det = a * e - b * d
    ia, ib, id_, ie = e / det, -b / det, -d / det, a / det
    return np.array([[ia, ib, -(ia * c + ib * f)], [id_, ie, -(id_ * c + ie * f)]], np.float64)


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
    trans = input_transform(c, s, inp_width, inp_height)
    resized = image if (new_height, new_width) == (height, width) else \
        resize_bilinear_u8(image, (new_width, new_height))
    inp = warp_bilinear_u8(resized, invert2x3(trans), (inp_width, inp_height))
    out = np.zeros((3, inp_height, inp_width), np.float32)
    for ch in range(3):
        m, sd = float(np.float32(mean[ch])), float(np.float32(std[ch]))
        out[ch] = ((inp[:, :, ch].astype(np.float64) / 255.) - m) / sd
    images = out[None]
    if flip_test:
        images = np.concatenate((images, images[:, :, :, ::-1]), axis=0)
    meta = {'c': c, 's': s, 'out_height': inp_height // down_ratio, 'out_width': inp_width // down_ratio}
    return np.ascontiguousarray(images), meta
