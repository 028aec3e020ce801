"""Repeatability probe of the deformable team kernel: the same launch N times, bitwise comparison (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_dcn as T
from oracle import cref
dev = torch.device("cuda:0")
for (Cin, HW, Cout) in [(128, 64, 64), (64, 128, 64), (256, 32, 128)]:
    B = 32
    x, off, mask, w, b = T._case(B, Cin, HW, HW, Cout, 300 + Cin)
    want = {i: cref.dcn_v2_forward(x[i:i + 1], off[i:i + 1], mask[i:i + 1], w, b) for i in (0, 13, 31)}
    first = None
    for it in range(int(os.environ.get("N", "12"))):
        for msig in (False, True):
            y = T._dcn_f32s_nhwc(dev, x, off, mask, w, b, 0, False, form=int(os.environ.get("FORM", "0")), msig=msig)
            bad = []
            for i in (0, 13, 31):
                err = np.abs(y[i:i + 1] - want[i]) / (1 + np.abs(want[i]))
                if err.max() >= 2e-5:
                    idx = np.unravel_index(np.argmax(err), err.shape)
                    nb = int((err >= 2e-5).sum())
                    bad.append((i, float(err.max()), idx, nb))
            if not msig:
                if first is None:
                    first = y
                elif not np.array_equal(first, y):
                    d = np.argwhere(first != y)
                    print("  NOT REPEATABLE", (Cin, HW, Cout), it, "cells", len(d), "first", d[:3].tolist())
            print((Cin, HW, Cout), "iter", it, "msig", msig, "bad", bad)
