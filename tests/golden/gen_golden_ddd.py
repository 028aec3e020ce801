#!/usr/bin/env python
"""Golden vectors for ddd_decode / _transpose_and_gather_feat, produced by RUNNING the
reference's src/lib/models/decode.py and models/utils.py on CPU tensors.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_ddd.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

from centernet_amd import synth  # noqa: E402

DDD_CASES = {
    # name: (B, C, H, W, K, wh, reg)      (kitti: 3 classes, 96 x 320 output, K = 40... scaled down)
    "ddd_kitti": (2, 3, 96, 320, 40, True, True),
    "ddd_plain": (1, 3, 24, 40, 12, False, False),
    "ddd_odd": (3, 2, 17, 23, 9, False, True),
}


def ddd_inputs(name):
    B, C, H, W, K, use_wh, use_reg = DDD_CASES[name]
    seed = 2000 + sum(map(ord, name))
    heat = synth.heatmap((B, C, H, W), seed)
    rot = synth.normal((B, 8, H, W), 1.0, seed + 1)
    depth = synth.uniform((B, 1, H, W), 1.0, 60.0, seed + 2)
    dim = synth.uniform((B, 3, H, W), 0.5, 4.0, seed + 3)
    wh = synth.uniform((B, 2, H, W), 0.0, 50.0, seed + 4) if use_wh else None
    reg = synth.uniform((B, 2, H, W), 0.0, 1.0, seed + 5) if use_reg else None
    return heat, rot, depth, dim, wh, reg, K


def main():
    import torch
    sys.path.insert(0, "/root/reference/src/lib")
    import models.decode as ref_decode            # namespace package: no model.py import
    from models.utils import _transpose_and_gather_feat

    def t(a):
        return None if a is None else torch.from_numpy(a.copy())
    out = {}
    for name in DDD_CASES:
        heat, rot, depth, dim, wh, reg, K = ddd_inputs(name)
        with torch.no_grad():
            dets = ref_decode.ddd_decode(t(heat), t(rot), t(depth), t(dim), wh=t(wh), reg=t(reg), K=K)
            s, i, c, y, x = ref_decode._topk(ref_decode._nms(t(heat)), K=K)
            g = _transpose_and_gather_feat(t(rot), i)
        out[name + "/dets"] = dets.numpy()
        out[name + "/inds"] = i.numpy().astype(np.int64)
        out[name + "/gather_rot"] = g.numpy()
    np.savez_compressed(os.path.join(HERE, "ddd_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
