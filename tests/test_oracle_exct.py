"""oracle exct_decode vs goldens produced by the reference's decode.py (CPU).  Rows are compared
bit-exactly; where the reference's top-k holds exactly equal scores (order unspecified by
torch.topk) the rows of such a run are compared as a set."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import cref

HERE = os.path.dirname(os.path.abspath(__file__))


def load_gen():
    spec = importlib.util.spec_from_file_location("gen_golden_exct",
                                                  os.path.join(HERE, "golden", "gen_golden_exct.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


GEN = load_gen()
GOLD = np.load(os.path.join(HERE, "golden", "exct_golden.npz"))


def canon(dets):
    """Sort rows inside runs of bit-equal scores (tie order is unspecified)."""
    out = dets.copy()
    for b in range(out.shape[0]):
        s = out[b, :, 4]
        start = 0
        for i in range(1, len(s) + 1):
            if i == len(s) or s[i] != s[start]:
                if i - start > 1:
                    blk = out[b, start:i]
                    out[b, start:i] = blk[np.lexsort(blk.T[::-1])]
                start = i
    return out


def assert_same(dets, ref):
    assert dets.shape == ref.shape
    assert np.array_equal(dets[..., 4].view(np.uint32), ref[..., 4].view(np.uint32))   # scores, in order
    a, r = canon(dets), canon(ref)
    for b in range(ref.shape[0]):
        n = ref.shape[1]
        # a tie run cut by the num_dets boundary may legitimately keep different members
        last = n
        while last > 0 and ref[b, last - 1, 4] == ref[b, n - 1, 4]:
            last -= 1
        assert np.array_equal(a[b, :last].view(np.uint32), r[b, :last].view(np.uint32))


@pytest.mark.parametrize("name", sorted(GEN.EXCT_CASES))
def test_oracle_exct_decode(name):
    heats, regs, K, num_dets = GEN.exct_inputs(name)
    dets = cref.exct_decode(*heats, *regs, K=K, num_dets=num_dets)
    ref = GOLD[name + "/dets"]
    assert int((ref[..., 4] > 0).sum()) > 0, "golden must contain valid groupings"
    assert_same(dets, ref)


@pytest.mark.parametrize("name", sorted(GEN.AGGR_CASES))
def test_oracle_edge_aggregation_and_decode(name):
    """aggr_weight > 0 (models/decode.py:17-90,136-140): the oracle's _h / _v aggregates are
    bit-identical to the reference's, and exct_decode behind them matches the reference's rows."""
    base, w = GEN.AGGR_CASES[name]
    heats, regs, K, num_dets = GEN.exct_inputs(base)
    assert np.array_equal(cref.h_aggregate(heats[0], w).view(np.uint32), GOLD[name + "/h_aggr"].view(np.uint32))
    assert np.array_equal(cref.v_aggregate(heats[1], w).view(np.uint32), GOLD[name + "/v_aggr"].view(np.uint32))
    dets = cref.exct_decode(*heats, *regs, K=K, num_dets=num_dets, aggr_weight=w)
    assert_same(dets, GOLD[name + "/dets"])
