#!/usr/bin/env python
"""Golden vectors for agnex_ct_decode (--agnostic_ex), produced by RUNNING the reference's
src/lib/models/decode.py:121-271 on CPU tensors.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_agnex.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

from centernet_amd import synth  # noqa: E402

AGNEX_CASES = {
    # name: (B, classes of the centre map, H, W, K, num_dets, regr, aggr_weight)
    "agnex_small": (2, 5, 24, 32, 6, 60, True, 0.0),
    "agnex_noregr": (1, 3, 16, 16, 5, 40, False, 0.0),
    "agnex_k40": (1, 80, 64, 64, 40, 1000, True, 0.0),
}


def agnex_inputs(name):
    """four (B, 1, H, W) edge maps + the (B, C, H, W) centre map (post-sigmoid), four offset maps or None"""
    B, C, H, W, K, num_dets, use_regr, aggr = AGNEX_CASES[name]
    seed = 7000 + sum(map(ord, name))
    rng = np.random.RandomState(seed)
    edges = [np.ascontiguousarray(synth.heatmap((B, 1, H, W), seed + e) * 0.3, dtype=np.float32) for e in range(4)]
    ct = np.ascontiguousarray(synth.heatmap((B, C, H, W), seed + 4) * 0.3, dtype=np.float32)
    for b in range(B):              # planted objects: four extreme points + a centre of some class
        for _ in range(3):
            c = rng.randint(C)
            x0, x1 = sorted(rng.choice(np.arange(1, W - 1), 2, replace=False))
            y0, y1 = sorted(rng.choice(np.arange(1, H - 1), 2, replace=False))
            tx, bx = rng.randint(x0, x1 + 1, 2)
            ly, ry = rng.randint(y0, y1 + 1, 2)
            edges[0][b, 0, y0, tx] = rng.uniform(0.6, 0.95)
            edges[1][b, 0, ly, x0] = rng.uniform(0.6, 0.95)
            edges[2][b, 0, y1, bx] = rng.uniform(0.6, 0.95)
            edges[3][b, 0, ry, x1] = rng.uniform(0.6, 0.95)
            ct[b, c, int((y0 + y1 + 0.5) / 2), int((x0 + x1 + 0.5) / 2)] = rng.uniform(0.5, 0.95)
    regs = [synth.uniform((B, 2, H, W), 0.0, 1.0, seed + 10 + e) if use_regr else None for e in range(4)]
    return edges + [ct], regs, K, num_dets


def main():
    import torch
    sys.path.insert(0, "/root/reference/src/lib")
    import models.decode as ref_decode            # namespace package: no model.py import

    def t(a):
        return None if a is None else torch.from_numpy(a.copy())
    out = {}
    for name in AGNEX_CASES:
        heats, regs, K, num_dets = agnex_inputs(name)
        with torch.no_grad():
            dets = ref_decode.agnex_ct_decode(*[t(h) for h in heats], *[t(r) for r in regs], K=K, num_dets=num_dets)
        out[name + "/dets"] = dets.numpy()
        print(name, dets.shape, "valid:", int((dets[..., 4] > 0).sum()), "classes:", sorted(set(dets[..., 13].reshape(-1).tolist()))[:8])
    np.savez_compressed(os.path.join(HERE, "agnex_golden.npz"), **out)


if __name__ == "__main__":
    main()
