#!/usr/bin/env python
"""Golden vectors for exct_decode, produced by RUNNING the reference's
src/lib/models/decode.py:273-424 on CPU tensors.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_exct.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

from centernet_amd import synth  # noqa: E402

EXCT_CASES = {
    # name: (B, C, H, W, K, num_dets, regr)
    "exct_small": (2, 3, 24, 32, 6, 60, True),
    "exct_noregr": (1, 2, 16, 16, 5, 40, False),
    "exct_k40": (1, 4, 64, 64, 40, 1000, True),
}


# edge aggregation (decode.py:17-90, aggr_weight > 0): name -> (base case, aggr_weight)
AGGR_CASES = {
    "exct_aggr": ("exct_small", 0.1),
    "exct_aggr_k40": ("exct_k40", 0.25),
    # aggregated peaks well above 1: the clamp behind the peak test (decode.py:302-305) decides scores,
    # ranks and the scores_thresh rule.  Weights chosen so that every (image, edge map) holds FEWER than K
    # clamped peaks: a run of scores == 1 that the K boundary cuts would make the reference's result depend
    # on torch.topk's unspecified tie order
    "exct_aggr_hot": ("exct_small", 2.5),
    "exct_aggr_hot_k40": ("exct_k40", 3.0),
}


def exct_inputs(name):
    B, C, H, W, K, num_dets, use_regr = EXCT_CASES[name]
    seed = 3000 + sum(map(ord, name))
    rng = np.random.RandomState(seed)
    # background: weak noise; planted objects: four extreme points + centre with strong scores,
    # so that some of the K^4 groupings pass every geometric / class / threshold rule
    heats = [np.ascontiguousarray(synth.heatmap((B, C, H, W), seed + e) * 0.3, dtype=np.float32)
             for e in range(5)]
    for b in range(B):
        for _ in range(3):
            c = rng.randint(C)
            x0, x1 = sorted(rng.choice(np.arange(1, W - 1), 2, replace=False))
            y0, y1 = sorted(rng.choice(np.arange(1, H - 1), 2, replace=False))
            tx, bx = rng.randint(x0, x1 + 1, 2)
            ly, ry = rng.randint(y0, y1 + 1, 2)
            pts = [(y0, tx), (ly, x0), (y1, bx), (ry, x1)]        # t, l, b, r
            for e, (y, x) in enumerate(pts):
                heats[e][b, c, y, x] = np.float32(rng.uniform(0.5, 0.95))
            cx, cy = int((x0 + x1 + 0.5) / 2), int((y0 + y1 + 0.5) / 2)
            heats[4][b, c, cy, cx] = np.float32(rng.uniform(0.4, 0.9))
    regs = [synth.uniform((B, 2, H, W), 0.0, 1.0, seed + 10 + e) if use_regr else None for e in range(4)]
    return heats, regs, K, num_dets


def main():
    import torch
    sys.path.insert(0, "/root/reference/src/lib")
    import models.decode as ref_decode

    def t(a):
        return None if a is None else torch.from_numpy(a.copy())
    out = {}
    for name in EXCT_CASES:
        heats, regs, K, num_dets = exct_inputs(name)
        with torch.no_grad():
            dets = ref_decode.exct_decode(*[t(h) for h in heats], *[t(r) for r in regs], K=K,
                                          num_dets=num_dets)
        out[name + "/dets"] = dets.numpy()
        print(name, dets.shape, "valid:", int((dets[..., 4] > 0).sum()))
    for name, (base, w) in AGGR_CASES.items():
        heats, regs, K, num_dets = exct_inputs(base)
        with torch.no_grad():
            # decode.py:136-140 by the reference's own functions, then the rest of exct_decode.  (Called
            # in one piece with aggr_weight > 0 the reference fails under torch >= 1.x: _v_aggregate
            # returns a transposed view and _topk's .view() needs contiguous memory; the reference's
            # pinned torch 0.4.1 produced contiguous results there.  The .contiguous() is the only
            # difference.)
            agg = [ref_decode._h_aggregate(t(heats[0]), aggr_weight=w).contiguous(),
                   ref_decode._v_aggregate(t(heats[1]), aggr_weight=w).contiguous(),
                   ref_decode._h_aggregate(t(heats[2]), aggr_weight=w).contiguous(),
                   ref_decode._v_aggregate(t(heats[3]), aggr_weight=w).contiguous()]
            dets = ref_decode.exct_decode(*agg, t(heats[4]), *[t(r) for r in regs], K=K,
                                          num_dets=num_dets, aggr_weight=0.0)
            out[name + "/h_aggr"] = agg[0].numpy()
            out[name + "/v_aggr"] = agg[1].numpy()
        out[name + "/dets"] = dets.numpy()
        print(name, dets.shape, "valid:", int((dets[..., 4] > 0).sum()))
    np.savez_compressed(os.path.join(HERE, "exct_golden.npz"), **out)


if __name__ == "__main__":
    main()
