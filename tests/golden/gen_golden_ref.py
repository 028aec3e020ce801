"""Fixtures produced by oracle/_ref -- the reference's own DCNv2 sampling kernel
(DCNv2/src/cuda/dcn_v2_im2col_cuda.cu, host build) and soft-NMS (external/nms.pyx, cython) --
so the pin survives where /root/reference (and hence `make -C oracle _ref`) is not available.

    python tests/golden/gen_golden_ref.py        # needs oracle/_ref built -> ref_golden.npz

Inputs are regenerated from seeds by the tests (dcn_inputs / nms_inputs below); only outputs are
stored: the full forward output and every COL_STRIDE-th column entry of sample 0.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_golden.npz")
COL_STRIDE = 7

# name -> B, Cin, H, W, Cout, kernel, stride, pad, dil, dg, offset sigma (px) or "stress"
DCN_CASES = {
    "kat_shape":  dict(B=2, Cin=2, H=4, W=4, Cout=2, k=3, stride=1, pad=1, dil=1, dg=1, sig=1.0),
    "c64_dg2":    dict(B=2, Cin=64, H=16, W=20, Cout=64, k=3, stride=1, pad=1, dil=1, dg=2, sig=2.0),
    "c128_64":    dict(B=2, Cin=128, H=16, W=16, Cout=64, k=3, stride=1, pad=1, dil=1, dg=1, sig=2.0),
    "c512_256":   dict(B=1, Cin=512, H=8, W=8, Cout=256, k=3, stride=1, pad=1, dil=1, dg=1, sig=1.5),
    "c32_stress": dict(B=2, Cin=32, H=12, W=10, Cout=16, k=3, stride=1, pad=1, dil=1, dg=1, sig="stress"),
    "c6_s2":      dict(B=2, Cin=6, H=17, W=13, Cout=5, k=3, stride=2, pad=1, dil=1, dg=1, sig=2.0),
    "c6_d2":      dict(B=1, Cin=6, H=15, W=14, Cout=4, k=3, stride=1, pad=2, dil=2, dg=3, sig=2.0),
    "c7_k5":      dict(B=1, Cin=7, H=11, W=12, Cout=3, k=5, stride=1, pad=2, dil=1, dg=1, sig=1.0),
}

NMS_CASES = {
    "m0_5": (5, 0, dict(Nt=0.5, method=0, threshold=0.001)),
    "m1_5": (5, 1, dict(Nt=0.5, method=1, threshold=0.05)),
    "m2_5": (5, 2, dict(Nt=0.5, method=2, threshold=0.05)),
    "m2_5_default": (5, 3, dict(Nt=0.5, method=2, threshold=0.001)),
    "m0_39": (39, 4, dict(Nt=0.5, method=0, threshold=0.001)),
    "m2_39": (39, 5, dict(Nt=0.5, method=2, threshold=0.05)),
}


def dcn_inputs(c):
    rs = np.random.RandomState(c["Cin"] * 31 + c["H"])
    k, s, p, d, dg = c["k"], c["stride"], c["pad"], c["dil"], c["dg"]
    Ho = (c["H"] + 2 * p - (d * (k - 1) + 1)) // s + 1
    Wo = (c["W"] + 2 * p - (d * (k - 1) + 1)) // s + 1
    x = rs.standard_normal((c["B"], c["Cin"], c["H"], c["W"])).astype(np.float32)
    if c["sig"] == "stress":
        off = rs.uniform(-c["H"], c["H"], (c["B"], dg * 2 * k * k, Ho, Wo)).astype(np.float32)
    else:
        off = (rs.standard_normal((c["B"], dg * 2 * k * k, Ho, Wo)) * c["sig"]).astype(np.float32)
    mask = rs.uniform(0, 1, (c["B"], dg * k * k, Ho, Wo)).astype(np.float32)
    w = (rs.standard_normal((c["Cout"], c["Cin"], k, k)) / np.sqrt(c["Cin"] * k * k)).astype(np.float32)
    b = rs.standard_normal(c["Cout"]).astype(np.float32)
    return x, off, mask, w, b, dict(stride=s, pad=p, dil=d, dg=dg)


def nms_inputs(ncol, seed, n=80):
    rs = np.random.RandomState(100 + seed)
    xy = rs.uniform(0, 120, (n, 2))
    wh = rs.uniform(10, 80, (n, 2))
    boxes = np.zeros((n, ncol), np.float32)
    boxes[:, 0:2] = xy
    boxes[:, 2:4] = xy + wh
    boxes[:, 4] = rs.uniform(0.0005, 1, n)
    if ncol > 5:
        boxes[:, 5:] = rs.uniform(0, 200, (n, ncol - 5))
    return boxes


def main():
    from oracle import ref
    assert ref.build(), "oracle/_ref could not be built (needs /root/reference)"
    out = {}
    for name, c in DCN_CASES.items():
        x, off, mask, w, b, kw = dcn_inputs(c)
        out["dcn_" + name + "_y"] = ref.dcn_v2_forward(x, off, mask, w, b, **kw)
        cols = ref.dcn_v2_im2col(x[0], off[0], mask[0], kh=c["k"], kw=c["k"], **kw)
        out["dcn_" + name + "_cols"] = cols.reshape(-1)[::COL_STRIDE].copy()
    for name, (ncol, seed, kw) in NMS_CASES.items():
        boxes = nms_inputs(ncol, seed)
        keep = (ref.soft_nms if ncol == 5 else ref.soft_nms_39)(boxes, **kw)
        out["nms_" + name + "_keep"] = np.int64(len(keep))
        out["nms_" + name + "_boxes"] = boxes
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
