"""Pre-process oracle (oracle/pre_oracle.py = OpenCV's published uint8 INTER_LINEAR algorithms,
restated) pinned by HAND-COMPUTED fixed-point cases, and the product's host path
(centernet_amd/image.py + the library's host routines) checked bit-exact against it.  CPU only.

Every expected number below is derived in the test from the published constants
(AB_BITS = 10, INTER_BITS = 5, INTER_REMAP_COEF_BITS = 15, INTER_RESIZE_COEF_BITS = 11) with
integer arithmetic one can redo on paper; where float bilinear would give a different uint8
level the comment says so -- those are the cases the round-2 float64 oracle got wrong."""
import numpy as np
import pytest

from centernet_amd import image as I
from oracle import pre_oracle as P


def _img(h, w, seed):
    return np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)


def _shift(tx, ty=0.0):
    """forward matrix dst = src + (tx, ty); warpAffine inverts it: src = dst - (tx, ty)."""
    return np.array([[1.0, 0.0, tx], [0.0, 1.0, ty]])


def test_weight_table():
    t = P.cv_bilinear_tab()
    assert t.shape == (1024, 4) and (t.sum(axis=1) == 32768).all() and t.min() >= 0 and t.max() <= 32767
    # fractions (fx, fy) in 1/32: weight = (32-fy)(32-fx)/1024 * 2^15 = (32-fy)(32-fx)*32, etc.
    assert t[0 * 32 + 22].tolist() == [10 * 32 * 32, 22 * 32 * 32, 0, 0]
    assert t[16 * 32 + 16].tolist() == [8192] * 4
    assert t[31 * 32 + 1].tolist() == [1 * 31 * 32, 1 * 1 * 32, 31 * 31 * 32, 31 * 1 * 32]
    # 1.0 * 32768 saturates at 32767 in int16; the table's sum fix-up puts the missing 1 on tap 3
    assert t[0].tolist() == [32767, 0, 0, 1]
    # ... which never shows: (32767*a + b + 16384) >> 15 == a for all uint8 a, b
    a, b = np.meshgrid(np.arange(256), np.arange(256))
    assert ((32767 * a + b + 16384) >> 15 == a).all()


def test_identity_and_integer_shift_copy_pixels():
    img = _img(12, 17, 0)
    assert np.array_equal(P.cv_warp_affine_u8(img, _shift(0.0), (17, 12)), img)
    sh = P.cv_warp_affine_u8(img, _shift(-3.0, 2.0), (17, 12))      # dst(x, y) = src(x + 3, y - 2)
    ref = np.zeros_like(img)
    ref[2:, :14] = img[:10, 3:]
    assert np.array_equal(sh, ref)


def test_half_pixel_rounds_half_up():
    """src = dst - 0.5:  X0 = rn(-0.5 * 1024) + 16 = -496;  X = (1024 x - 496) >> 5 = 32 x - 16
    (arithmetic shift floors -15.5 to -16): tap x - 1, fraction 16/32, weights 16384 / 16384:
    dst = (16384 (p[x-1] + p[x]) + 16384) >> 15 = (p[x-1] + p[x] + 1) >> 1 -- half goes UP
    (numpy.rint / float bilinear + half-even would turn 10.5 into 10)."""
    row = np.array([[10, 11, 20, 25, 0, 255]], np.uint8)
    got = P.cv_warp_affine_u8(row, _shift(0.5), (6, 1))[0].tolist()
    assert got == [(0 + 10 + 1) >> 1, (10 + 11 + 1) >> 1, (11 + 20 + 1) >> 1, (20 + 25 + 1) >> 1,
                   (25 + 0 + 1) >> 1, (0 + 255 + 1) >> 1]
    assert got == [5, 11, 16, 23, 13, 128]
    # both axes: fractions (16, 16), four weights of 8192: (a + b + c + d + 2) >> 2
    im = np.array([[10, 13], [20, 24]], np.uint8)
    assert P.cv_warp_affine_u8(im, _shift(0.5, 0.5), (2, 2))[1, 1] == (10 + 13 + 20 + 24 + 2) >> 2 == 17


def test_sample_positions_are_quantised_to_one_32nd_pixel():
    """src = dst - 0.3:  X0 = rn(-307.2) + 16 = -291;  X = (1024 x - 291) >> 5 = 32 x - 10
    (-291 / 32 = -9.09 floors to -10): tap x - 1, fraction 22/32 = 0.6875 -- not 0.7 -- weights
    10240 / 22528.  p = (200, 100): (200*10240 + 100*22528 + 16384) >> 15 = 4317184 >> 15 = 131;
    float bilinear gives 200*0.3 + 100*0.7 = 130."""
    row = np.array([[200, 100, 50, 255, 0, 7]], np.uint8)
    got = P.cv_warp_affine_u8(row, _shift(0.3), (6, 1))[0].tolist()
    p = [0] + row[0].tolist()
    assert got == [(p[i] * 10240 + p[i + 1] * 22528 + 16384) >> 15 for i in range(6)]
    assert got == [138, 131, 66, 191, 80, 5]
    assert got[1] == 131 and round(200 * 0.3 + 100 * 0.7) == 130


def test_coordinate_rounding_constant():
    """round_delta = 1024 / 32 / 2 = 16 rounds the position to the NEAREST 1/32 pixel:
    src = dst + 15/1024: X0 = 15 + 16 = 31 -> 31 >> 5 = 0 -> fraction 0 (exact copy);
    src = dst + 16/1024 (1/64 px, the midpoint): X0 = 32 -> 1 -> fraction 1/32:
    (a*31*32*32 + b*1*32*32 + 16384) >> 15."""
    row = np.array([[64, 192, 0, 255]], np.uint8)
    assert P.cv_warp_affine_u8(row, _shift(-15.0 / 1024), (4, 1))[0].tolist() == [64, 192, 0, 255]
    got = P.cv_warp_affine_u8(row, _shift(-16.0 / 1024), (4, 1))[0].tolist()
    q = row[0].tolist() + [0]
    assert got == [(q[i] * 31744 + q[i + 1] * 1024 + 16384) >> 15 for i in range(4)] == [68, 186, 8, 247]
    # row term and column term are rounded separately: X = (rn((m1 y + m2) 1024) + 16 + rn(m0 x 1024)) >> 5
    M = np.array([[1.0004, 0.0, 0.0], [0.0, 1.0, 0.0]])           # inverse m0 = 1/1.0004
    m0 = 1.0 / 1.0004
    xs = np.arange(40)
    X = (16 + np.rint((m0 * xs) * 1024).astype(np.int64)) >> 5
    ramp = (np.arange(40) * 6).astype(np.uint8)[None]
    got = P.cv_warp_affine_u8(ramp, M, (40, 1))[0]
    sx, fx = X >> 5, X & 31
    r = np.concatenate([ramp[0].astype(np.int64), [0]])
    want = (r[sx] * (32 - fx) * 1024 + r[sx + 1] * fx * 1024 + 16384) >> 15
    want[fx == 0] = r[sx[fx == 0]]
    assert got.tolist() == want.tolist()


def test_zero_border_and_far_outside():
    im = np.full((2, 2), 200, np.uint8)
    # src = dst - 0.5 in x and y: pixel (0, 0) mixes one image tap (weight 8192) with three zeros
    assert P.cv_warp_affine_u8(im, _shift(0.5, 0.5), (3, 3)).tolist() == \
        [[(200 * 8192 + 16384) >> 15, (2 * 200 * 8192 + 16384) >> 15, (200 * 8192 + 16384) >> 15],
         [(2 * 200 * 8192 + 16384) >> 15, 200, (2 * 200 * 8192 + 16384) >> 15],
         [(200 * 8192 + 16384) >> 15, (2 * 200 * 8192 + 16384) >> 15, (200 * 8192 + 16384) >> 15]]
    assert P.cv_warp_affine_u8(im, _shift(1e4, -1e4), (3, 3)).max() == 0


def test_resize_same_size_is_a_copy_and_half_size_is_the_area_mean():
    img = _img(6, 8, 1)
    assert np.array_equal(P.cv_resize_linear_u8(img, (8, 6)), img)
    half = P.cv_resize_linear_u8(img, (4, 3)).astype(int)
    s = img.astype(int)
    assert np.array_equal(half, (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2)


def test_resize_times_two_by_hand():
    """2 x 2 -> 4 x 4, scale = 0.5 per axis.
    columns: dx = 0: f = 0.25 - 0.5 < 0 -> clamped, (tap 0, 2048, 0); dx = 1: f = 0.25 -> (0, 1536, 512);
             dx = 2: f = 0.75 -> (0, 512, 1536); dx = 3: f = 1.25 -> tap 1 is the last -> (1, 2048, 0).
    rows:    dy = 0: f = -0.25 -> tap -1, fraction 0.75 kept: (512, 1536) on rows clip(-1) = 0, clip(0) = 0;
             dy = 1: (0: 1536, 1: 512); dy = 2: (0: 512, 1: 1536); dy = 3: tap 1, f = 0.25: rows 1, clip(2) = 1.
    vertical pass: (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2."""
    im = np.array([[11, 50], [90, 133]], np.uint8)
    H = [[11 * 2048, 11 * 1536 + 50 * 512, 11 * 512 + 50 * 1536, 50 * 2048],
         [90 * 2048, 90 * 1536 + 133 * 512, 90 * 512 + 133 * 1536, 133 * 2048]]
    rows = [(0, 0, 512, 1536), (0, 1, 1536, 512), (0, 1, 512, 1536), (1, 1, 1536, 512)]
    want = [[(((b0 * (H[r0][x] >> 4)) >> 16) + ((b1 * (H[r1][x] >> 4)) >> 16) + 2) >> 2 for x in range(4)]
            for (r0, r1, b0, b1) in rows]
    assert want == [[11, 21, 40, 50], [31, 41, 61, 71], [70, 81, 102, 112], [90, 101, 122, 133]]
    assert P.cv_resize_linear_u8(im, (4, 4)).tolist() == want
    # (1, 1): float bilinear = 0.75 (0.75*11 + 0.25*50) + 0.25 (0.75*90 + 0.25*133) = 40.56 -> 41: agrees;
    # (2, 1): 0.25*20.75 + 0.75*100.75 = 80.75 -> 81: agrees; the truncating shifts differ elsewhere:
    im2 = np.array([[0, 100, 255]], np.uint8)
    # 3 -> 2 columns (scale 1.5): dx = 0: f = 0.25, (0, 1536, 512): S = 51200; dx = 1: f = 1.75 -> tap 1,
    # f = 0.75, (512, 1536): S = 100*512 + 255*1536 = 442880; one row, f = 0: b = (2048, 0):
    # ((2048 * (51200 >> 4)) >> 16 = 100, + 2) >> 2 = 25;  ((2048 * 27680) >> 16 = 865, + 2) >> 2 = 216
    assert P.cv_resize_linear_u8(im2, (2, 1)).tolist() == [[25, 216]]


def test_affine_through_three_points():
    # the benchmark geometry (512 x 512 frame, fix_res): exactly the identity
    t = P.get_affine_transform(np.array([256., 256.], np.float32), 512.0, 0, [512, 512])
    assert np.array_equal(t, np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]))
    # 640 x 480 frame into 512 x 512: scale 0.8 about the centres
    t = P.get_affine_transform(np.array([320., 240.], np.float32), 640.0, 0, [512, 512])
    np.testing.assert_allclose(t, [[0.8, 0.0, 0.0], [0.0, 0.8, 64.0]], rtol=0, atol=1e-12)
    # the product's two 3 x 3 eliminations == the literal 6 x 6 elimination, bit for bit
    rs = np.random.RandomState(0)
    for _ in range(300):
        c = np.array([rs.uniform(10, 700), rs.uniform(10, 500)], np.float32)
        s = float(rs.uniform(50, 900))
        rot = float(rs.choice([0, 0, 30, -12.5, 90, rs.uniform(-180, 180)]))
        size = (int(rs.choice([512, 384, 64])), int(rs.choice([512, 96])))
        for inv in (0, 1):
            assert np.array_equal(I.get_affine_transform(c, s, rot, size, inv=inv),
                                  P.get_affine_transform(c, s, rot, list(size), inv=inv))
    for _ in range(300):
        src = rs.uniform(-100, 100, (3, 2)).astype(np.float32)
        dst = rs.uniform(-100, 100, (3, 2)).astype(np.float32)
        if rs.rand() < 0.4:
            src[1, 0] = src[0, 0]          # pivot ties: the row order of the 6 x 6 system decides
        assert np.array_equal(I.affine_through(src, dst), P.cv_get_affine_transform(src, dst))
    t = np.array([[0.8, 0.1, 3.0], [-0.2, 1.1, -7.5]])
    assert np.array_equal(I.invert_affine(t), P.cv_invert_affine(t))


def test_host_routines_equal_the_oracle_bit_for_bit():
    rs = np.random.RandomState(1)
    img = _img(97, 131, 2)
    for _ in range(30):
        c = np.array([rs.uniform(20, 110), rs.uniform(20, 80)], np.float32)
        size = (int(rs.choice([64, 96, 128])), int(rs.choice([64, 96])))
        M = I.get_affine_transform(c, float(rs.uniform(40, 260)), float(rs.choice([0, 0, 17.0])), size)
        assert np.array_equal(I.warp_affine(img, M, size), P.cv_warp_affine_u8(img, M, size))
    for size in [(200, 150), (40, 33), (131, 97), (65, 48), (262, 194), (100, 97), (131, 50)]:
        assert np.array_equal(I.resize_bilinear(img, size), P.cv_resize_linear_u8(img, size))
    even = _img(96, 130, 3)
    assert np.array_equal(I.resize_bilinear(even, (65, 48)), P.cv_resize_linear_u8(even, (65, 48)))
    gray = img[:, :, 0].copy()
    assert np.array_equal(I.resize_bilinear(gray, (75, 96)), P.cv_resize_linear_u8(gray, (75, 96)))
    assert np.array_equal(I.warp_affine(gray, _shift(0.3, -1.7), (80, 70)),
                          P.cv_warp_affine_u8(gray, _shift(0.3, -1.7), (80, 70)))


@pytest.mark.parametrize("shape,scale,fix", [((64, 64), 1.0, True), ((37, 53), 1.0, True),
                                             ((40, 30), 1.0, False), ((32, 48), 0.75, True),
                                             ((24, 20), 1.5, False), ((48, 40), 0.5, True),
                                             ((480, 640), 1.0, True)])
def test_host_pre_process_matches_oracle(shape, scale, fix):
    """BaseDetector.pre_process (host form) == the oracle's base_detector.py:37-65, bit for bit:
    matrix, resize, warp, normalisation, flip -- fix_res / keep_res, several scales."""
    from centernet_amd.opts import opts
    from centernet_amd.detectors.base_detector import BaseDetector
    res = "512" if shape[0] > 100 else "64"
    args = ["ctdet", "--input_h", res, "--input_w", res, "--gpus", "-1", "--flip_test"]
    if not fix:
        args.append("--keep_res")
    opt = opts().init(args)
    det = BaseDetector.__new__(BaseDetector)
    det.opt = opt
    det.mean = np.array(opt.mean, dtype=np.float32).reshape(1, 1, 3)
    det.std = np.array(opt.std, dtype=np.float32).reshape(1, 1, 3)
    img = _img(shape[0], shape[1], 7)
    images, meta = det.pre_process(img, scale)
    ref, rmeta = P.pre_process(img, scale, opt.mean, opt.std, fix_res=opt.fix_res, input_h=int(res),
                               input_w=int(res), pad=opt.pad, flip_test=True, down_ratio=opt.down_ratio)
    assert images.shape == ref.shape
    assert np.array_equal(images.numpy().view(np.uint32), ref.view(np.uint32))
    assert meta['out_height'] == rmeta['out_height'] and meta['out_width'] == rmeta['out_width']
    assert np.array_equal(meta['c'], rmeta['c']) and np.array_equal(np.asarray(meta['s']), np.asarray(rmeta['s']))
