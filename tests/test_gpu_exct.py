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


@pytest.mark.parametrize("name", sorted(GEN.AGGR_CASES))
def test_exct_decode_with_edge_aggregation(dev, name):
    """aggr_weight > 0 (models/decode.py:17-90,136-140): cn_exct_aggregate_f32 bit-identical to the
    reference's _h_aggregate / _v_aggregate, the decode behind it to the reference's rows and, bit for
    bit, to the oracle."""
    base, w = GEN.AGGR_CASES[name]
    heats, regs, K, num_dets = GEN.exct_inputs(base)
    lib = native.lib()
    B, C, H, W = heats[0].shape
    for which, horizontal, key in ((0, 1, "/h_aggr"), (1, 0, "/v_aggr")):
        x = _t(heats[which], dev)
        o = torch.empty_like(x)
        native.check(lib.cn_exct_aggregate_f32(native.ptr(x), native.ptr(o), B, C, H, W, horizontal, w,
                                               native.stream_ptr()), "cn_exct_aggregate_f32")
        assert np.array_equal(o.cpu().numpy().view(np.uint32), GOLD[name + key].view(np.uint32))
    dets = D.exct_decode(*[_t(h, dev) for h in heats], *[_t(r, dev) for r in regs], K=K,
                         num_dets=num_dets, aggr_weight=w).cpu().numpy()
    assert_same(dets, GOLD[name + "/dets"])
    ref = cref.exct_decode(*heats, *regs, K=K, num_dets=num_dets, aggr_weight=w)
    assert np.array_equal(dets.view(np.uint32), ref.view(np.uint32))
