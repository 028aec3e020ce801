"""Bindings of oracle/_ref/ -- the REFERENCE's own native code compiled test-only
(TEST INFRASTRUCTURE, see oracle/__init__.py and oracle/Makefile target `_ref`).

    dcn_v2_im2col / dcn_v2_forward  <- modulated_deformable_im2col_cuda
                                       (DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:314-337; kernel
                                       :118-180, bilinear :18-47) driven like dcn_v2_cuda_forward
                                       (DCNv2/src/dcn_v2_cuda.c:61-97) by dcn_v2_ref_driver.cpp
    soft_nms / soft_nms_39          <- external/nms.pyx:77-275 (cython)

`available()` is False when the libraries were never built (no /root/reference at build time);
tests skip in that case and fall back on the committed fixtures made from these libraries
(tests/golden/gen_golden_ref.py).
"""
import ctypes
import glob
import importlib.util
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "_ref")
_DCN_SO = os.path.join(_DIR, "libdcn_v2_ref.so")
REFERENCE_ROOT = "/root/reference"
_f32p = ctypes.POINTER(ctypes.c_float)
_dcn = None
_nms = None


def _nms_path():
    hits = sorted(glob.glob(os.path.join(_DIR, "ref_nms*.so")))
    return hits[0] if hits else None


def build():
    """`make -C oracle _ref` when the reference tree is present; otherwise keep what is there."""
    if os.path.isdir(REFERENCE_ROOT):
        subprocess.check_call(["make", "-C", _HERE, "_ref"], stdout=subprocess.DEVNULL)
    return available()


def available():
    return os.path.exists(_DCN_SO) and _nms_path() is not None


def _dcn_lib():
    global _dcn
    if _dcn is None:
        _dcn = ctypes.CDLL(_DCN_SO)
        _dcn.ref_dcn_v2_forward.restype = ctypes.c_int
        _dcn.ref_dcn_v2_im2col.restype = ctypes.c_int
    return _dcn


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _out_hw(H, W, kh, kw, stride, pad, dil):
    return ((H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1,
            (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1)


def dcn_v2_im2col(x, offset, mask, kh=3, kw=3, stride=1, pad=1, dil=1, dg=1):
    """One sample through the reference kernel: x (Cin,H,W) -> columns (Cin*kh*kw, Ho, Wo)."""
    x, xp = _f(x)
    offset, op = _f(offset)
    mask, mp = _f(mask)
    Cin, H, W = x.shape
    Ho, Wo = _out_hw(H, W, kh, kw, stride, pad, dil)
    assert offset.shape == (dg * 2 * kh * kw, Ho, Wo), offset.shape
    assert mask.shape == (dg * kh * kw, Ho, Wo), mask.shape
    cols = np.empty((Cin * kh * kw, Ho, Wo), np.float32)
    rc = _dcn_lib().ref_dcn_v2_im2col(xp, op, mp, cols.ctypes.data_as(_f32p), Cin, H, W, kh, kw,
                                      stride, stride, pad, pad, dil, dil, dg)
    if rc != 0:
        raise RuntimeError("ref_dcn_v2_im2col failed: %d" % rc)
    return cols


def dcn_v2_forward(x, offset, mask, weight, bias, stride=1, pad=1, dil=1, dg=1):
    x, xp = _f(x)
    offset, op = _f(offset)
    mask, mp = _f(mask)
    weight, wp = _f(weight)
    bias, bp = _f(bias)
    B, Cin, H, W = x.shape
    Cout, Cin_w, kh, kw = weight.shape
    if Cin_w != Cin:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (Cin, Cin_w))
    Ho, Wo = _out_hw(H, W, kh, kw, stride, pad, dil)
    assert offset.shape == (B, dg * 2 * kh * kw, Ho, Wo), offset.shape
    assert mask.shape == (B, dg * kh * kw, Ho, Wo), mask.shape
    out = np.empty((B, Cout, Ho, Wo), np.float32)
    rc = _dcn_lib().ref_dcn_v2_forward(xp, wp, bp, op, mp, out.ctypes.data_as(_f32p), B, Cin, H, W,
                                       Cout, kh, kw, stride, stride, pad, pad, dil, dil, dg)
    if rc != 0:
        raise RuntimeError("ref_dcn_v2_forward failed: %d" % rc)
    return out


def nms_module():
    """The reference's cython module (soft_nms, soft_nms_39, soft_nms_merge, nms)."""
    global _nms
    if _nms is None:
        spec = importlib.util.spec_from_file_location("ref_nms", _nms_path())
        _nms = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_nms)
    return _nms


def soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """external/nms.pyx:77-170: modifies `boxes` in place (as the reference does), returns keep."""
    return nms_module().soft_nms(boxes, sigma=sigma, Nt=Nt, threshold=threshold, method=method)


def soft_nms_39(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """external/nms.pyx:172-275."""
    return nms_module().soft_nms_39(boxes, sigma=sigma, Nt=Nt, threshold=threshold, method=method)
