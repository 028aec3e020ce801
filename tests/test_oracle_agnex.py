"""oracle agnex_ct_decode (the class-agnostic ExtremeNet grouping behind --agnostic_ex) vs goldens produced by
the reference's models/decode.py:121-271 on CPU; rows bit-exact, runs of exactly equal scores as sets."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import cref
from test_oracle_exct import assert_same

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("gen_golden_agnex", os.path.join(HERE, "golden", "gen_golden_agnex.py"))
GEN = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(GEN)
GOLD = np.load(os.path.join(HERE, "golden", "agnex_golden.npz"))


@pytest.mark.parametrize("name", sorted(GEN.AGNEX_CASES))
def test_oracle_agnex_ct_decode(name):
    heats, regs, K, num_dets = GEN.agnex_inputs(name)
    dets = cref.agnex_ct_decode(*heats, *regs, K=K, num_dets=num_dets)
    ref = GOLD[name + "/dets"]
    assert int((ref[..., 4] > 0).sum()) > 0, "golden must contain valid groupings"
    assert len(set(ref[..., 13].reshape(-1).tolist())) > 1, "golden must carry more than one class"
    assert_same(dets, ref)


def test_agnostic_form_is_the_class_form_on_single_channel_maps_plus_the_argmax_class():
    """What the device entry relies on: scores and geometry of agnex_ct_decode == exct_decode over the
    single-channel edge maps against the per-cell maximum of the centre map; only the class column differs."""
    heats, regs, K, num_dets = GEN.agnex_inputs("agnex_small")
    agn = cref.agnex_ct_decode(*heats, *regs, K=K, num_dets=num_dets)
    ct_max = heats[4].max(axis=1, keepdims=True)
    cls = cref.exct_decode(*heats[:4], ct_max, *regs, K=K, num_dets=num_dets)
    assert np.array_equal(agn[..., :13].view(np.uint32), cls[..., :13].view(np.uint32))
    assert np.all(cls[..., 13] == 0) and agn[..., 13].max() > 0
    # the class is the arg-max of the centre map at the integer box centre of the un-offset points
    B, C, H, W = heats[4].shape
    assert np.all((agn[..., 13] >= 0) & (agn[..., 13] < C))
