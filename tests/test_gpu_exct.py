"""exct_decode on the GPU (through the C ABI) vs the reference goldens and the oracle."""
import numpy as np
import pytest
import torch

from centernet_amd import decode as D, native
from oracle import cref
from test_oracle_exct import GEN, GOLD, assert_same

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return None if a is None else torch.from_numpy(a).to(dev)


@pytest.mark.parametrize("name", sorted(GEN.EXCT_CASES))
def test_exct_decode_matches_reference_golden(dev, name):
    heats, regs, K, num_dets = GEN.exct_inputs(name)
    dets = D.exct_decode(*[_t(h, dev) for h in heats], *[_t(r, dev) for r in regs], K=K,
                         num_dets=num_dets).cpu().numpy()
    assert_same(dets, GOLD[name + "/dets"])
    # the device's own tie rule (candidate index ascending) is the oracle's: fully bit-exact
    ref = cref.exct_decode(*heats, *regs, K=K, num_dets=num_dets)
    assert np.array_equal(dets.view(np.uint32), ref.view(np.uint32))


def test_exct_decode_partial_regr_and_errors(dev):
    heats, regs, K, num_dets = GEN.exct_inputs("exct_small")
    # decode.py:372-373: regression is applied only when all four maps are given
    some = [regs[0], None, regs[2], regs[3]]
    got = D.exct_decode(*[_t(h, dev) for h in heats], *[_t(r, dev) for r in some], K=K,
                        num_dets=num_dets).cpu().numpy()
    ref = cref.exct_decode(*heats, None, None, None, None, K=K, num_dets=num_dets)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    with pytest.raises(RuntimeError):
        D.exct_decode(*[_t(h, dev) for h in heats], K=K, num_dets=K ** 4 + 1)
    with pytest.raises(native.NativeError):
        D.exct_decode(*[_t(h, dev) for h in heats], K=K, num_dets=10, aggr_weight=0.1)
