"""Pre-process oracle (oracle/pre_oracle.py) pinned by closed-form cases, and the host
restatement (centernet_amd/image.py) checked bit-exact against it.  CPU only."""
import numpy as np
import pytest

from centernet_amd import image as I
from oracle import pre_oracle as P


def _img(h, w, seed):
    return np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)


def test_identity_and_integer_shift():
    img = _img(12, 17, 0)
    eye = [1, 0, 0, 0, 1, 0]
    assert np.array_equal(P.warp_bilinear_u8(img, eye, (17, 12)), img)
    sh = P.warp_bilinear_u8(img, [1, 0, 3, 0, 1, -2], (17, 12))   # dst(x,y) = src(x+3, y-2)
    ref = np.zeros_like(img)
    ref[2:, :14] = img[:10, 3:]
    assert np.array_equal(sh, ref)


def test_half_pixel_taps_and_rounding():
    img = np.zeros((2, 2, 3), np.uint8)
    img[0, 0], img[0, 1], img[1, 0], img[1, 1] = 10, 13, 20, 24
    # (0.5, 0): mean(10, 13) = 11.5 -> 12 (half to even); (0.5, 0.5): 16.75 -> 17
    assert P.warp_bilinear_u8(img, [1, 0, 0.5, 0, 1, 0], (1, 1))[0, 0, 0] == 12
    assert P.warp_bilinear_u8(img, [1, 0, 0.5, 0, 1, 0.5], (1, 1))[0, 0, 0] == 17
    img[0, 1] = 11                                              # mean(10, 11) = 10.5 -> 10
    assert P.warp_bilinear_u8(img, [1, 0, 0.5, 0, 1, 0], (1, 1))[0, 0, 0] == 10
    # zero border: sampling at x = -0.5 mixes the border pixel with 0
    assert P.warp_bilinear_u8(img, [1, 0, -0.5, 0, 1, 0], (1, 1))[0, 0, 0] == 5
    # replicate border (cv2.resize): same position returns the edge pixel
    assert P.warp_bilinear_u8(img, [1, 0, -0.5, 0, 1, 0], (1, 1), replicate=True)[0, 0, 0] == 10


def test_resize_exact_cases():
    img = _img(6, 8, 1)
    assert np.array_equal(P.resize_bilinear_u8(img, (8, 6)), img)
    up = P.resize_bilinear_u8(img, (16, 12))                    # x2: taps at 0.25 / 0.75
    v = 0.75 * (0.75 * float(img[0, 0, 0]) + 0.25 * float(img[0, 1, 0])) + \
        0.25 * (0.75 * float(img[1, 0, 0]) + 0.25 * float(img[1, 1, 0]))
    assert abs(int(up[1, 1, 0]) - v) <= 0.5
    assert np.array_equal(up[0, 0], img[0, 0])                  # corner: replicated border


def test_affine_solve_matches_numpy():
    c = np.array([33.5, 20.0], np.float32)
    t = P.input_transform(c, 67.0, 64, 48)
    t2 = I.get_affine_transform(c, 67.0, 0, [64, 48])
    np.testing.assert_allclose(t, t2, rtol=0, atol=1e-12)
    np.testing.assert_allclose(P.invert2x3(t), I.invert_affine(t2), rtol=0, atol=1e-12)


@pytest.mark.parametrize("shape,scale,fix", [((64, 64), 1.0, True), ((37, 53), 1.0, True),
                                             ((40, 30), 1.0, False), ((32, 48), 0.75, True),
                                             ((24, 20), 1.5, False)])
def test_host_pre_process_matches_oracle(shape, scale, fix):
    """centernet_amd's vectorised host path == scalar oracle, bit for bit (same float64
    operation order), for fix_res / keep_res, scales, flip."""
    from centernet_amd.opts import opts
    from centernet_amd.detectors.base_detector import BaseDetector
    args = ["ctdet", "--input_h", "64", "--input_w", "64", "--gpus", "-1", "--flip_test"]
    if not fix:
        args.append("--keep_res")
    opt = opts().init(args)
    det = BaseDetector.__new__(BaseDetector)
    det.opt = opt
    det.mean = np.array(opt.mean, dtype=np.float32).reshape(1, 1, 3)
    det.std = np.array(opt.std, dtype=np.float32).reshape(1, 1, 3)
    img = _img(shape[0], shape[1], 7)
    images, meta = det.pre_process(img, scale)
    ref, rmeta = P.pre_process(img, scale, opt.mean, opt.std, fix_res=opt.fix_res, input_h=64,
                               input_w=64, pad=opt.pad, flip_test=True, down_ratio=opt.down_ratio)
    assert images.shape == ref.shape
    # (a) end to end: the two affine solves (numpy LU vs Cramer) may differ in the last ulp of
    # the matrix, which can move a sample across a rounding boundary on rare pixels: at most
    # one uint8 level, on < 1 % of the pixels
    diff = np.abs(images.numpy() - ref)
    lvl = (1.0 / 255.0) / float(min(opt.std))
    assert float(diff.max()) <= lvl * 1.0001
    assert float((diff > 0).mean()) < 1e-2
    assert meta['out_height'] == rmeta['out_height'] and np.allclose(meta['c'], rmeta['c'])
    # (b) same matrix in both: bit-identical warp + normalise
    h, w = img.shape[:2]
    nh, nw = int(h * scale), int(w * scale)
    ih, iw = images.shape[2], images.shape[3]
    trans = I.get_affine_transform(meta['c'], meta['s'], 0, [iw, ih])
    Mi = I.invert_affine(trans)
    res_h = I.resize_bilinear(img, (nw, nh))
    res_o = img if (nh, nw) == (h, w) else P.resize_bilinear_u8(img, (nw, nh))
    assert np.array_equal(res_h, res_o)
    wo = P.warp_bilinear_u8(res_o, Mi, (iw, ih))
    assert np.array_equal(I.warp_bilinear_u8(res_h, Mi, (iw, ih)), wo)
    assert np.array_equal(I.normalize_chw(wo, opt.mean, opt.std), images.numpy()[0])
