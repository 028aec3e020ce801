"""soft_nms / soft_nms_39 (mirror of src/lib/external/nms.pyx:77-275): in-place on a float32
host array, returns the list of kept row indices.  Native (C) implementation in
libcenternet_amd.so (``cn_soft_nms_f32``)."""
import ctypes

import numpy as np

from . import native


def _run(boxes, stride, sigma, Nt, threshold, method):
    if boxes.dtype != np.float32 or boxes.ndim != 2 or boxes.shape[1] != stride or \
            not boxes.flags["C_CONTIGUOUS"]:
        raise ValueError("boxes must be a C-contiguous float32 (N,%d) array" % stride)
    n = native.lib().cn_soft_nms_f32(boxes.ctypes.data_as(ctypes.c_void_p), boxes.shape[0], stride,
                                     float(sigma), float(Nt), float(threshold), int(method))
    if n < 0:
        raise native.NativeError("cn_soft_nms_f32 failed (%d)" % n)
    return list(range(n))


def soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    return _run(boxes, 5, sigma, Nt, threshold, method)


def soft_nms_39(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    return _run(boxes, 39, sigma, Nt, threshold, method)
