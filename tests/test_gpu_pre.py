"""Device pre-process (cn_resize_bilinear_u8 / cn_warp_normalize_u8_f32, through the C ABI)
vs oracle/pre_oracle.py -- OpenCV's fixed-point INTER_LINEAR restated -- bit-exact (uint8 levels
and fp32 bits), and the detector's pre_process_device vs its host pre_process."""
import ctypes

import numpy as np
import pytest
import torch

from centernet_amd import native, image as I
from oracle import pre_oracle as P

pytestmark = pytest.mark.gpu
MEAN = [0.408, 0.447, 0.470]
STD = [0.289, 0.274, 0.278]


def _img(h, w, seed):
    return np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)


def _warp_norm(dev, img, Mi, oh, ow, flip):
    lib = native.lib()
    src = torch.from_numpy(img).to(dev)
    out = torch.full((2 if flip else 1, 3, oh, ow), float("nan"), device=dev)
    mi = (ctypes.c_double * 6)(*np.asarray(Mi, np.float64).reshape(-1))
    mean, std = (ctypes.c_float * 3)(*MEAN), (ctypes.c_float * 3)(*STD)
    native.check(lib.cn_warp_normalize_u8_f32(native.ptr(src), img.shape[0], img.shape[1],
                                              img.shape[1] * 3, mi, oh, ow, mean, std, int(flip),
                                              native.ptr(out), native.stream_ptr()), "warp")
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("case", [
    # (H, W), out (h, w), dst->src matrix
    ((24, 31), (24, 31), [1, 0, 0, 0, 1, 0]),                       # identity: exact copy
    ((24, 31), (32, 32), [1, 0, 3, 0, 1, -2]),                      # integer shift + zero border
    ((20, 28), (32, 40), [0.7, 0.05, -1.3, -0.04, 0.66, 2.2]),      # general affine
    ((33, 17), (16, 48), [1.9, 0, -20.5, 0, 2.3, -4.25]),           # mostly outside the image
    ((8, 8), (8, 8), [1, 0, 1e5, 0, 1, -1e5]),                      # far outside: all zero
])
@pytest.mark.parametrize("flip", [False, True])
def test_warp_normalize_bit_exact(dev, case, flip):
    (h, w), (oh, ow), m = case
    img = _img(h, w, 3)
    got = _warp_norm(dev, img, m, oh, ow, flip)
    u8 = P.cv_warp_affine_u8(img, np.array(m, np.float64).reshape(2, 3), (ow, oh), inverse_map=True)
    ref = I.normalize_chw(u8, MEAN, STD)[None]
    if flip:
        ref = np.concatenate((ref, ref[:, :, :, ::-1]), axis=0)
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(ref).view(np.uint32))


@pytest.mark.parametrize("shape,out", [((12, 16), (24, 32)), ((40, 30), (17, 23)), ((9, 9), (9, 9)),
                                       ((31, 64), (48, 23)), ((40, 30), (20, 15)), ((375, 500), (281, 375)),
                                       ((6, 7), (1, 1)), ((1, 1), (5, 3))])
def test_resize_bit_exact(dev, shape, out):
    lib = native.lib()
    img = _img(shape[0], shape[1], 5)
    src = torch.from_numpy(img).to(dev)
    dst = torch.zeros((out[0], out[1], 3), device=dev, dtype=torch.uint8)
    native.check(lib.cn_resize_bilinear_u8(native.ptr(src), shape[0], shape[1], shape[1] * 3,
                                           out[0], out[1], native.ptr(dst), native.stream_ptr()),
                 "resize")
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), P.cv_resize_linear_u8(img, (out[1], out[0])))


@pytest.mark.parametrize("shape,scale,keep,flip", [((512, 512), 1.0, False, False),
                                                   ((375, 500), 1.0, False, True),
                                                   ((120, 90), 1.5, True, False),
                                                   ((200, 333), 0.5, False, True)])
def test_detector_pre_process_device_equals_host(dev, shape, scale, keep, flip):
    from centernet_amd.opts import opts
    from centernet_amd.detectors.base_detector import BaseDetector
    args = ["ctdet"] + (["--keep_res"] if keep else []) + (["--flip_test"] if flip else [])
    opt = opts().init(args)
    opt.device = dev
    det = BaseDetector.__new__(BaseDetector)
    det.opt = opt
    det.mean = np.array(opt.mean, dtype=np.float32).reshape(1, 1, 3)
    det.std = np.array(opt.std, dtype=np.float32).reshape(1, 1, 3)
    img = _img(shape[0], shape[1], 11)
    a, ma = det.pre_process(img, scale)
    b, mb = det.pre_process_device(img, scale)
    torch.cuda.synchronize()
    assert b.is_cuda and tuple(a.shape) == tuple(b.shape)
    assert np.array_equal(a.numpy().view(np.uint32), b.cpu().numpy().view(np.uint32))
    assert ma['out_height'] == mb['out_height'] and ma['out_width'] == mb['out_width']
    assert np.array_equal(ma['c'], mb['c']) and np.array_equal(np.asarray(ma['s']), np.asarray(mb['s']))


def test_warp_random_sweep_bit_exact(dev):
    """40 seeded random (size, matrix, flip) cases: device == scalar oracle, bit for bit."""
    rng = np.random.RandomState(2024)
    for case in range(40):
        h, w = int(rng.randint(1, 40)), int(rng.randint(1, 40))
        oh, ow = int(rng.randint(1, 48)), int(rng.randint(1, 48))
        m = [rng.uniform(-2, 2), rng.uniform(-1, 1), rng.uniform(-20, 20),
             rng.uniform(-1, 1), rng.uniform(-2, 2), rng.uniform(-20, 20)]
        if case % 5 == 0:   # exact half-pixel positions: the rounding rule matters
            m = [1.0, 0.0, float(rng.randint(-3, 3)) + 0.5, 0.0, 1.0, float(rng.randint(-3, 3)) + 0.5]
        img = _img(h, w, 100 + case)
        flip = bool(case & 1)
        got = _warp_norm(dev, img, m, oh, ow, flip)
        ref = I.normalize_chw(P.cv_warp_affine_u8(img, np.array(m, np.float64).reshape(2, 3), (ow, oh), inverse_map=True), MEAN, STD)[None]
        if flip:
            ref = np.concatenate((ref, ref[:, :, :, ::-1]), axis=0)
        assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(ref).view(np.uint32)), case
