"""oracle ddd_decode / gather vs goldens produced by the reference's decode.py (CPU, bit-exact)."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import cref

HERE = os.path.dirname(os.path.abspath(__file__))


def load_gen():
    spec = importlib.util.spec_from_file_location("gen_golden_ddd",
                                                  os.path.join(HERE, "golden", "gen_golden_ddd.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


GEN = load_gen()
GOLD = np.load(os.path.join(HERE, "golden", "ddd_golden.npz"))


@pytest.mark.parametrize("name", sorted(GEN.DDD_CASES))
def test_oracle_ddd_decode_bit_exact(name):
    heat, rot, depth, dim, wh, reg, K = GEN.ddd_inputs(name)
    dets = cref.ddd_decode(heat, rot, depth, dim, wh=wh, reg=reg, K=K)
    ref = GOLD[name + "/dets"]
    assert dets.shape == ref.shape
    assert np.array_equal(dets.view(np.uint32), ref.view(np.uint32))
    g = cref.transpose_and_gather_feat(rot, GOLD[name + "/inds"])
    assert np.array_equal(g, GOLD[name + "/gather_rot"])
