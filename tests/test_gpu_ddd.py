"""ddd_decode / _topk / _transpose_and_gather_feat on the GPU (through the C ABI) vs the
reference goldens and the oracle: bit-exact."""
import os

import numpy as np
import pytest
import torch

from centernet_amd import decode as D
from oracle import cref
from test_oracle_ddd import GEN, GOLD

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return None if a is None else torch.from_numpy(a).to(dev)


@pytest.mark.parametrize("name", sorted(GEN.DDD_CASES))
def test_ddd_decode_matches_reference_golden(dev, name):
    heat, rot, depth, dim, wh, reg, K = GEN.ddd_inputs(name)
    dets = D.ddd_decode(_t(heat, dev), _t(rot, dev), _t(depth, dev), _t(dim, dev), wh=_t(wh, dev),
                        reg=_t(reg, dev), K=K).cpu().numpy()
    ref = GOLD[name + "/dets"]
    assert np.array_equal(dets.view(np.uint32), ref.view(np.uint32))
    s, i, c, y, x = D._topk(_t(heat, dev), K=K, nms=True)
    assert np.array_equal(i.cpu().numpy(), GOLD[name + "/inds"])
    g = D._transpose_and_gather_feat(_t(rot, dev), i)
    assert np.array_equal(g.cpu().numpy(), GOLD[name + "/gather_rot"])
    so, io, co, yo, xo = cref.topk(cref.nms(heat), K)
    assert np.array_equal(s.cpu().numpy(), so) and np.array_equal(c.cpu().numpy(), co)
    assert np.array_equal(y.cpu().numpy(), yo) and np.array_equal(x.cpu().numpy(), xo)


def test_ddd_decode_fused_sigmoid(dev):
    """apply_sigmoid=True (logits in): scores within 1e-6 of torch's sigmoid; every other column
    identical wherever the oracle's own neighbouring scores differ by more than 1e-6."""
    heat, rot, depth, dim, wh, reg, K = GEN.ddd_inputs("ddd_odd")
    logits = np.log(heat / (1 - heat)).astype(np.float32)
    hs = torch.from_numpy(logits.copy()).sigmoid().numpy()
    ref = cref.ddd_decode(hs, rot, depth, dim, wh=wh, reg=reg, K=K)
    got = D.ddd_decode(_t(logits, dev), _t(rot, dev), _t(depth, dev), _t(dim, dev), wh=_t(wh, dev),
                       reg=_t(reg, dev), K=K, apply_sigmoid=True).cpu().numpy()
    assert np.abs(got[..., 2] - ref[..., 2]).max() < 1e-6
    sc = ref[..., 2]
    gap = np.minimum(np.abs(np.diff(sc, axis=1, prepend=np.inf)), np.abs(np.diff(sc, axis=1, append=-np.inf)))
    safe = gap > 1e-6
    assert safe.mean() > 0.9
    cols = [c for c in range(ref.shape[2]) if c != 2]
    assert np.array_equal(got[safe][:, cols], ref[safe][:, cols])


def test_ddd_decode_errors(dev):
    heat, rot, depth, dim, wh, reg, K = GEN.ddd_inputs("ddd_plain")
    with pytest.raises(RuntimeError):
        D.ddd_decode(_t(heat, dev), _t(rot, dev), _t(depth, dev), _t(dim, dev), K=24 * 40 + 1)
