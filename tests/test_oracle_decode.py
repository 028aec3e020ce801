"""oracle/decode_oracle.c against the golden vectors produced by the reference's own
models/decode.py (tests/golden/gen_golden.py).  Bit-exact."""
import numpy as np
import pytest

from oracle import cref


def test_golden_inputs_unchanged(gen, decode_golden):
    """The seeded input builders still produce the bytes the goldens were made from."""
    _, meta = decode_golden
    for name in gen.DECODE_CASES:
        heat, wh, reg, K, cat = gen.decode_inputs(name)
        assert gen.sha(heat, wh, reg) == meta[name]["sha"], name
    for name in gen.POSE_CASES:
        assert gen.sha(*gen.pose_inputs(name)[:6]) == meta[name]["sha"], name


@pytest.mark.parametrize("name", ["ctdet_coco", "ctdet_small_catspec", "ctdet_rect", "ctdet_odd"])
def test_ctdet_decode_matches_reference(gen, decode_golden, name):
    z, meta = decode_golden
    heat, wh, reg, K, cat = gen.decode_inputs(name)
    assert meta[name]["min_gap"] > 0, "golden case has a score tie inside the top-K"
    dets, inds = cref.ctdet_decode(heat, wh, reg, cat_spec_wh=cat, K=K, return_inds=True)
    ref = z[name + "/dets"]
    assert dets.shape == ref.shape
    assert np.array_equal(dets.view(np.uint32), ref.view(np.uint32)), \
        "max abs diff %g" % np.abs(dets - ref).max()
    assert np.array_equal(inds, z[name + "/topk_inds"])


@pytest.mark.parametrize("name", ["ctdet_coco", "ctdet_small_catspec", "ctdet_rect", "ctdet_odd"])
def test_topk_matches_reference(gen, decode_golden, name):
    z, _ = decode_golden
    heat, wh, reg, K, cat = gen.decode_inputs(name)
    nmsd = cref.nms(heat)
    if (name + "/nms") in z.files:
        assert np.array_equal(nmsd, z[name + "/nms"])
    s, i, c, y, x = cref.topk(nmsd, K)
    assert np.array_equal(s, z[name + "/topk_score"])
    assert np.array_equal(i, z[name + "/topk_inds"])
    assert np.array_equal(c, z[name + "/topk_clses"])
    assert np.array_equal(y, z[name + "/topk_ys"])
    assert np.array_equal(x, z[name + "/topk_xs"])
    cs, ci, cy, cx = cref.topk_channel(nmsd, K)
    ref_s, ref_i = z[name + "/chan_score"], z[name + "/chan_inds"]
    assert np.array_equal(cs, ref_s)
    # per-channel lists may contain ties (suppressed cells are all 0): compare indices
    # only where the score is strictly separated from its neighbours
    strict = np.ones_like(ref_s, dtype=bool)
    strict[..., 1:] &= ref_s[..., 1:] < ref_s[..., :-1]
    strict[..., :-1] &= ref_s[..., :-1] > ref_s[..., 1:]
    assert np.array_equal(ci[strict], ref_i[strict])


@pytest.mark.parametrize("name", ["pose_full", "pose_no_hm_hp", "pose_no_offsets"])
def test_multi_pose_decode_matches_reference(gen, decode_golden, name):
    z, _ = decode_golden
    heat, wh, kps, reg, hm_hp, hp_offset, K = gen.pose_inputs(name)
    dets = cref.multi_pose_decode(heat, wh, kps, reg, hm_hp, hp_offset, K)
    ref = z[name + "/dets"]
    assert dets.shape == ref.shape
    bad = dets.view(np.uint32) != ref.view(np.uint32)
    assert not bad.any(), "%d mismatches, max abs %g" % (bad.sum(), np.abs(dets - ref).max())


def test_tie_rule_is_score_class_index():
    """All-equal heat-map: every cell is a peak; order must be class asc, index asc."""
    heat = np.full((1, 3, 4, 5), 0.25, np.float32)
    wh = np.zeros((1, 2, 4, 5), np.float32)
    dets, inds = cref.ctdet_decode(heat, wh, None, K=7, return_inds=True)
    assert list(inds[0]) == [0, 1, 2, 3, 4, 5, 6]
    assert np.all(dets[0, :, 5] == 0)
    assert np.all(dets[0, :, 4] == 0.25)


def test_k_larger_than_map_raises():
    heat = np.zeros((1, 1, 2, 2), np.float32)
    with pytest.raises(RuntimeError):
        cref.ctdet_decode(heat, np.zeros((1, 2, 2, 2), np.float32), None, K=5)
