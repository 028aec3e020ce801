"""HIP decode (through the C ABI) vs the reference goldens and the C oracle."""
import numpy as np
import pytest
import torch

from centernet_amd import synth
from oracle import cref

pytestmark = pytest.mark.gpu


def _gpu(a, dev):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("name", ["ctdet_coco", "ctdet_small_catspec", "ctdet_rect", "ctdet_odd"])
def test_ctdet_decode_bit_exact_vs_reference_golden(dev, gen, decode_golden, name):
    from centernet_amd.decode import ctdet_decode
    z, _ = decode_golden
    heat, wh, reg, K, cat = gen.decode_inputs(name)
    dets, inds = ctdet_decode(_gpu(heat, dev), _gpu(wh, dev), _gpu(reg, dev), cat_spec_wh=cat, K=K,
                              return_inds=True)
    dets, inds = dets.cpu().numpy(), inds.cpu().numpy()
    ref = z[name + "/dets"]
    assert np.array_equal(inds, z[name + "/topk_inds"]), "box indices differ from the reference"
    assert np.array_equal(dets.view(np.uint32), ref.view(np.uint32)), np.abs(dets - ref).max()


@pytest.mark.parametrize("shape", [(1, 80, 128, 128, 100), (4, 80, 128, 128, 100), (32, 80, 128, 128, 100),
                                   (2, 1, 128, 128, 100), (1, 3, 152, 100, 40), (3, 2, 7, 9, 5),
                                   (1, 2, 10, 13, 128), (2, 20, 96, 320, 100)])
def test_ctdet_decode_vs_oracle(dev, shape):
    from centernet_amd.decode import ctdet_decode
    B, C, H, W, K = shape
    heat = synth.heatmap((B, C, H, W), 11 + B + C)
    wh = synth.uniform((B, 2, H, W), 0, 40, 5)
    reg = synth.uniform((B, 2, H, W), 0, 1, 6)
    ref, ref_inds = cref.ctdet_decode(heat, wh, reg, K=K, return_inds=True)
    dets, inds = ctdet_decode(_gpu(heat, dev), _gpu(wh, dev), _gpu(reg, dev), K=K, return_inds=True)
    assert np.array_equal(inds.cpu().numpy(), ref_inds)
    assert np.array_equal(dets.cpu().numpy().view(np.uint32), ref.view(np.uint32))


def test_fused_sigmoid_matches_oracle(dev):
    """apply_sigmoid=True (logits in): scores within 1e-6 of torch's sigmoid, same boxes
    wherever the oracle's own score gap exceeds 1e-6."""
    from centernet_amd.decode import ctdet_decode
    B, C, H, W, K = 4, 80, 128, 128, 100
    logits = synth.normal((B, C, H, W), 2.0, 3, mean=-2.19)
    wh = synth.uniform((B, 2, H, W), 0, 40, 5)
    reg = synth.uniform((B, 2, H, W), 0, 1, 6)
    heat = torch.from_numpy(logits.copy()).sigmoid_().numpy()
    ref, ref_inds = cref.ctdet_decode(heat, wh, reg, K=K, return_inds=True)
    dets, inds = ctdet_decode(_gpu(logits, dev), _gpu(wh, dev), _gpu(reg, dev), K=K,
                              apply_sigmoid=True, return_inds=True)
    dets, inds = dets.cpu().numpy(), inds.cpu().numpy()
    assert np.abs(dets[..., 4] - ref[..., 4]).max() < 1e-6
    from oracle.parity import compare_topk
    ids = np.stack([inds, dets[..., 5].astype(np.int64)], -1)
    rids = np.stack([ref_inds, ref[..., 5].astype(np.int64)], -1)
    r = compare_topk(dets, ref, tie=1e-6, got_ids=ids, ref_ids=rids)
    assert r["paired"] >= 0.999 and r["in_place"] >= 0.995 and r["safe"] >= 0.9, r   # (the per-rank rules are asserted inside compare_topk)


def test_ties_and_degenerate_maps(dev):
    from centernet_amd.decode import ctdet_decode
    # constant map: every cell is a peak and all scores tie -> (class, index) order
    heat = np.full((2, 3, 16, 16), 0.25, np.float32)
    wh = synth.uniform((2, 2, 16, 16), 0, 4, 1)
    ref, ri = cref.ctdet_decode(heat, wh, None, K=100, return_inds=True)
    d, i = ctdet_decode(_gpu(heat, dev), _gpu(wh, dev), None, K=100, return_inds=True)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # fewer than K peaks: a single bump per plane, the rest of the top-K are zeros
    heat = np.zeros((1, 2, 12, 12), np.float32)
    heat[0, 0, 5, 6] = 0.9
    heat[0, 1, 2, 3] = 0.8
    heat[0, 1, 2, 4] = 0.8   # equal neighbours: both survive the == test
    ref, ri = cref.ctdet_decode(heat, wh[:1, :, :12, :12].copy(), None, K=20, return_inds=True)
    d, i = ctdet_decode(_gpu(heat, dev), _gpu(wh[:1, :, :12, :12].copy(), dev), None, K=20,
                        return_inds=True)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # negative "heat" (the function accepts any floats)
    heat = synth.normal((1, 4, 9, 9), 1.0, 2)
    ref, ri = cref.ctdet_decode(heat, wh[:1, :, :9, :9].copy(), None, K=30, return_inds=True)
    d, i = ctdet_decode(_gpu(heat, dev), _gpu(wh[:1, :, :9, :9].copy(), dev), None, K=30,
                        return_inds=True)
    assert np.array_equal(d.cpu().numpy()[..., 4], ref[..., 4])
    assert np.array_equal(i.cpu().numpy(), ri)


def test_k_out_of_range_and_cpu_tensor_raise(dev):
    from centernet_amd.decode import ctdet_decode
    from centernet_amd.native import NativeError
    heat = torch.zeros((1, 1, 2, 2), device=dev)
    with pytest.raises(RuntimeError):
        ctdet_decode(heat, torch.zeros((1, 2, 2, 2), device=dev), K=5)
    with pytest.raises(NativeError):
        ctdet_decode(torch.zeros((1, 1, 8, 8)), torch.zeros((1, 2, 8, 8)), K=5)


def test_full_size_properties(dev):
    """BASELINE config size (B=32): size-independent properties -- sorted scores, every
    box is a true local maximum, idempotent under re-decode, permutation of the batch."""
    from centernet_amd.decode import ctdet_decode
    B, C, H, W, K = 32, 80, 128, 128, 100
    g = torch.Generator(device="cpu").manual_seed(0)
    logits = (2 * torch.randn((B, C, H, W), generator=g) - 2.19).to(dev)
    wh = (40 * torch.rand((B, 2, H, W), generator=g)).to(dev)
    reg = torch.rand((B, 2, H, W), generator=g).to(dev)
    dets, inds = ctdet_decode(logits, wh, reg, K=K, apply_sigmoid=True, return_inds=True)
    s = dets[..., 4]
    assert bool((s[:, 1:] <= s[:, :-1]).all())
    heat = logits.sigmoid()
    hmax = torch.nn.functional.max_pool2d(heat, 3, 1, 1)
    cls = dets[..., 5].long()
    b = torch.arange(B, device=dev)[:, None].expand(B, K)
    y, x = inds // W, inds % W
    assert bool((hmax[b, cls, y, x] == heat[b, cls, y, x]).all())
    assert float((heat[b, cls, y, x] - s).abs().max()) < 1e-6
    # the K-th score bounds every non-selected peak
    peaks = torch.where(hmax == heat, heat, torch.zeros_like(heat)).view(B, -1)
    kth = torch.topk(peaks, K, dim=1).values[:, -1]
    assert float((kth - s[:, -1]).abs().max()) < 1e-6
    perm = torch.randperm(B, generator=g).to(dev)
    d2 = ctdet_decode(logits[perm].contiguous(), wh[perm].contiguous(), reg[perm].contiguous(),
                      K=K, apply_sigmoid=True)
    assert torch.equal(d2, dets[perm])


@pytest.mark.parametrize("name", ["pose_full", "pose_no_hm_hp", "pose_no_offsets"])
def test_multi_pose_decode_bit_exact_vs_reference_golden(dev, gen, decode_golden, name):
    from centernet_amd.decode import multi_pose_decode
    z, _ = decode_golden
    heat, wh, kps, reg, hm_hp, hp_offset, K = gen.pose_inputs(name)
    dets = multi_pose_decode(_gpu(heat, dev), _gpu(wh, dev), _gpu(kps, dev), _gpu(reg, dev),
                             _gpu(hm_hp, dev), _gpu(hp_offset, dev), K=K).cpu().numpy()
    ref = z[name + "/dets"]
    bad = dets.view(np.uint32) != ref.view(np.uint32)
    assert not bad.any(), "%d mismatches, max abs %g" % (bad.sum(), np.abs(dets - ref).max())


@pytest.mark.parametrize("shape", [(8, 128, 128, 100), (1, 96, 160, 64), (3, 17, 23, 9)])
def test_multi_pose_decode_vs_oracle(dev, shape):
    from centernet_amd.decode import multi_pose_decode
    B, H, W, K = shape
    J = 17
    heat = synth.heatmap((B, 1, H, W), 3)
    wh = synth.uniform((B, 2, H, W), 0, 60, 4)
    kps = synth.normal((B, 2 * J, H, W), 8.0, 5)
    reg = synth.uniform((B, 2, H, W), 0, 1, 6)
    hm_hp = np.sqrt(np.sqrt(synth.heatmap((B, J, H, W), 7))).astype(np.float32) * synth.heatmap((B, J, H, W), 8)
    hp_offset = synth.uniform((B, 2, H, W), 0, 1, 9)
    ref = cref.multi_pose_decode(heat, wh, kps, reg, hm_hp, hp_offset, K)
    dets = multi_pose_decode(_gpu(heat, dev), _gpu(wh, dev), _gpu(kps, dev), _gpu(reg, dev),
                             _gpu(hm_hp, dev), _gpu(hp_offset, dev), K=K).cpu().numpy()
    bad = dets.view(np.uint32) != ref.view(np.uint32)
    assert not bad.any(), "%d mismatches, max abs %g" % (bad.sum(), np.abs(dets - ref).max())


def test_full_scan_fallback_equals_compacted_path(dev):
    """The kernel's exact fallback (taken for degenerate maps) and its compacted-peak fast
    path must agree bit for bit on ordinary data (debug flag 1024 forces the fallback)."""
    from centernet_amd.decode import ctdet_decode
    B, C, H, W, K = 3, 20, 128, 128, 100
    heat = torch.from_numpy(synth.heatmap((B, C, H, W), 5)).to(dev)
    wh = torch.from_numpy(synth.uniform((B, 2, H, W), 0, 40, 6)).to(dev)
    a, ia = ctdet_decode(heat, wh, None, K=K, return_inds=True)
    b, ib = ctdet_decode(heat, wh, None, K=K, return_inds=True, _debug_flags=1024)
    assert torch.equal(ia, ib) and torch.equal(a, b)


@pytest.mark.parametrize("name", ["ctdet_coco", "ctdet_small_catspec", "ctdet_rect", "ctdet_odd"])
def test_topk_channel_direct_vs_reference_golden(dev, gen, decode_golden, name):
    """cn_nms_topk_channel_f32 on its own against the reference's `_topk_channel(_nms(heat))`
    goldens (decode.py:92-101): scores bit-exact everywhere, indices wherever the score is
    strictly separated from its neighbours (suppressed cells all tie at 0)."""
    from centernet_amd.decode import _topk_channel
    z, _ = decode_golden
    heat, wh, reg, K, cat = gen.decode_inputs(name)
    s, i, ys, xs = _topk_channel(_gpu(heat, dev), K=K, nms=True)
    s, i, ys, xs = s.cpu().numpy(), i.cpu().numpy(), ys.cpu().numpy(), xs.cpu().numpy()
    ref_s, ref_i = z[name + "/chan_score"], z[name + "/chan_inds"]
    assert s.shape == ref_s.shape and i.shape == ref_i.shape
    assert np.array_equal(s.view(np.uint32), ref_s.view(np.uint32))
    strict = np.ones_like(ref_s, dtype=bool)
    strict[..., 1:] &= ref_s[..., 1:] < ref_s[..., :-1]
    strict[..., :-1] &= ref_s[..., :-1] > ref_s[..., 1:]
    assert strict.mean() > 0.5
    assert np.array_equal(i[strict], ref_i[strict])
    W = heat.shape[3]
    assert np.array_equal(ys, (i // W).astype(np.float32)) and np.array_equal(xs, (i % W).astype(np.float32))


@pytest.mark.parametrize("shape", [(2, 5, 24, 40, 30), (1, 17, 128, 128, 100), (3, 2, 7, 9, 5)])
def test_topk_plain_functions_without_nms(dev, shape):
    """_topk_channel / _topk as plain functions (decode.py:92-119, no _nms in front) on maps
    with negative values too, vs the C oracle."""
    from centernet_amd.decode import _topk_channel, _topk
    B, C, H, W, K = shape
    scores = synth.normal((B, C, H, W), 1.0, 21)
    rs, ri, ry, rx = cref.topk_channel(scores, K)
    s, i, ys, xs = _topk_channel(_gpu(scores, dev), K=K, nms=False)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), rs.view(np.uint32))
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(ys.cpu().numpy(), ry) and np.array_equal(xs.cpu().numpy(), rx)
    ts, ti, tc, ty, tx = cref.topk(scores, K)
    s, i, c, ys, xs = _topk(_gpu(scores, dev), K=K, nms=False)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), ts.view(np.uint32))
    assert np.array_equal(i.cpu().numpy(), ti) and np.array_equal(c.cpu().numpy(), tc)


def test_mismatched_inputs_raise_instead_of_reading_out_of_bounds(dev):
    """The wrappers pass raw pointers: shapes, channel counts and index ranges are validated
    first, as the reference's torch ops would raise."""
    from centernet_amd.decode import ctdet_decode, multi_pose_decode, _transpose_and_gather_feat
    heat = torch.rand((2, 3, 16, 16), device=dev)
    wh = torch.rand((2, 2, 16, 16), device=dev)
    with pytest.raises(RuntimeError):
        ctdet_decode(heat, wh[:, :, :8].contiguous(), K=10)            # wrong H
    with pytest.raises(RuntimeError):
        ctdet_decode(heat, wh, K=10, cat_spec_wh=True)                 # needs 2*C wh channels
    with pytest.raises(RuntimeError):
        ctdet_decode(heat, wh, reg=torch.rand((1, 2, 16, 16), device=dev), K=10)
    with pytest.raises(RuntimeError):
        multi_pose_decode(heat[:, :1].contiguous(), wh, torch.rand((2, 34, 16, 16), device=dev),
                          hm_hp=torch.rand((2, 16, 16, 16), device=dev), K=10)   # 16 != 17 joints
    feat = torch.rand((2, 4, 16, 16), device=dev)
    ok = _transpose_and_gather_feat(feat, torch.tensor([[0, 255], [17, 3]], device=dev))
    assert tuple(ok.shape) == (2, 2, 4)
    with pytest.raises(RuntimeError):
        _transpose_and_gather_feat(feat, torch.tensor([[0, 256], [1, 2]], device=dev))
    with pytest.raises(RuntimeError):
        _transpose_and_gather_feat(feat, torch.tensor([[0, -1], [1, 2]], device=dev))


@pytest.mark.parametrize("shape", [(4, 80, 128, 128, 100), (2, 100, 64, 64, 50), (3, 24, 96, 160, 40),
                                   (2, 40, 50, 70, 20)])
def test_image_level_select_equals_per_band_select(dev, shape):
    """The threshold-pruned image-level top-K (group maxima -> K-th largest as threshold ->
    candidate keys -> exact select) and the per-(class, band) select of round 1 (debug flag
    2048) give bit-identical detections; both equal the oracle."""
    from centernet_amd.decode import ctdet_decode
    B, C, H, W, K = shape
    heat = synth.heatmap((B, C, H, W), 31 + C)
    wh = synth.uniform((B, 2, H, W), 0, 40, 5)
    reg = synth.uniform((B, 2, H, W), 0, 1, 6)
    a, ia = ctdet_decode(_gpu(heat, dev), _gpu(wh, dev), _gpu(reg, dev), K=K, return_inds=True)
    b, ib = ctdet_decode(_gpu(heat, dev), _gpu(wh, dev), _gpu(reg, dev), K=K, return_inds=True,
                         _debug_flags=2048)
    assert torch.equal(ia, ib) and torch.equal(a, b)
    ref, ref_inds = cref.ctdet_decode(heat, wh, reg, K=K, return_inds=True)
    assert np.array_equal(ia.cpu().numpy(), ref_inds)
    assert np.array_equal(a.cpu().numpy().view(np.uint32), ref.view(np.uint32))


def test_image_level_select_degenerate_maps(dev):
    """Inputs on which the threshold is useless -- constant maps (every cell ties), fewer than
    K positive peaks, thousands of cells tying the K-th score -- take the exact full-scan path
    of the merge kernel: still bit-exact vs the oracle (ties: class asc, index asc)."""
    from centernet_amd.decode import ctdet_decode
    B, C, H, W, K = 2, 100, 64, 64, 50
    wh = synth.uniform((B, 2, H, W), 0, 40, 5)
    cases = []
    cases.append(np.full((B, C, H, W), 0.25, np.float32))                     # constant
    few = np.zeros((B, C, H, W), np.float32)                                   # 7 positive peaks
    for i in range(7):
        few[0, 3 * i, 5 + 2 * i, 9 + 3 * i] = 0.9 - 0.1 * i
    few[1, 99, 63, 63] = 0.5
    cases.append(few)
    plateau = synth.heatmap((B, C, H, W), 77)
    plateau[:, :, ::2, ::2] = np.maximum(plateau[:, :, ::2, ::2], 0.97)        # 100k cells at 0.97+
    plateau[:, :, ::2, ::2] = 0.97
    cases.append(plateau.astype(np.float32))
    for heat in cases:
        ref, ref_inds = cref.ctdet_decode(heat, wh, None, K=K, return_inds=True)
        d, i = ctdet_decode(_gpu(heat, dev), _gpu(wh, dev), None, K=K, return_inds=True)
        assert np.array_equal(i.cpu().numpy(), ref_inds)
        assert np.array_equal(d.cpu().numpy().view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("shape", [(4, 80, 128, 128, 100), (2, 100, 64, 64, 50), (2, 24, 112, 96, 40),
                                   (3, 30, 40, 128, 20), (1, 80, 128, 128, 128)])
@pytest.mark.parametrize("sig", [False, True])
def test_one_launch_form_equals_the_other_forms(dev, shape, sig):
    """Planes of <= 128 x 128 cells take the ONE-launch form (plane in registers, lane-maximum
    threshold, image floor, last-arriver select).  It must be bit-identical to the two-launch form
    (flag 8192) and the per-band select (2048), with the library's own fill of the state words
    (debug-flag calls share a scratch workspace) and without it (the default call owns a zeroed
    workspace, CN_DECODE_STATE_CLEAN); post-sigmoid input also against the oracle."""
    from centernet_amd.decode import ctdet_decode
    B, C, H, W, K = shape
    heat = synth.heatmap((B, C, H, W), 131 + C + H)
    if sig:
        heat = np.log(np.clip(heat, 1e-6, 1 - 1e-6) / (1 - np.clip(heat, 1e-6, 1 - 1e-6))).astype(np.float32)
    wh = synth.uniform((B, 2, H, W), 0, 40, 5)
    reg = synth.uniform((B, 2, H, W), 0, 1, 6)
    args = (_gpu(heat, dev), _gpu(wh, dev), _gpu(reg, dev))
    a, ia = ctdet_decode(*args, K=K, apply_sigmoid=sig, return_inds=True)                       # one launch, clean state
    f, i_f = ctdet_decode(*args, K=K, apply_sigmoid=sig, return_inds=True, _debug_flags=16384)   # one launch + fill
    t, it = ctdet_decode(*args, K=K, apply_sigmoid=sig, return_inds=True, _debug_flags=8192)
    p, ip_ = ctdet_decode(*args, K=K, apply_sigmoid=sig, return_inds=True, _debug_flags=2048)
    for d, i in ((f, i_f), (t, it), (p, ip_)):
        assert torch.equal(ia, i) and torch.equal(a, d)
    if not sig:
        ref, ref_inds = cref.ctdet_decode(heat, wh, reg, K=K, return_inds=True)
        assert np.array_equal(ia.cpu().numpy(), ref_inds)
        assert np.array_equal(a.cpu().numpy().view(np.uint32), ref.view(np.uint32))


def test_one_launch_form_leaves_its_state_clean(dev):
    """Back-to-back calls on the SAME owned workspace (no fill in between) with different data,
    a degenerate map among them: every call equals the oracle, i.e. the last arriver of every image
    left list length, arrival counter and floor at zero."""
    from centernet_amd.decode import ctdet_decode
    B, C, H, W, K = 3, 40, 128, 128, 100
    wh = synth.uniform((B, 2, H, W), 0, 40, 5)
    maps = [synth.heatmap((B, C, H, W), 7), np.full((B, C, H, W), 0.5, np.float32),
            synth.heatmap((B, C, H, W), 8), np.zeros((B, C, H, W), np.float32), synth.heatmap((B, C, H, W), 9)]
    for rep in range(2):
        for heat in maps:
            ref, ref_inds = cref.ctdet_decode(heat, wh, None, K=K, return_inds=True)
            d, i = ctdet_decode(_gpu(heat, dev), _gpu(wh, dev), None, K=K, return_inds=True)
            assert np.array_equal(i.cpu().numpy(), ref_inds)
            assert np.array_equal(d.cpu().numpy().view(np.uint32), ref.view(np.uint32))


def test_one_launch_form_plateaus_and_sparse_planes(dev):
    """Inputs on which the lane-maximum threshold is useless: saturated blocks (thousands of cells
    tie at the top score: the plane's list overflows LDS -> exact select over the register cells),
    planes with a handful of peaks (zeros of suppressed cells take part), a single hot plane."""
    from centernet_amd.decode import ctdet_decode
    B, C, H, W, K = 2, 80, 128, 128, 100
    wh = synth.uniform((B, 2, H, W), 0, 40, 5)
    reg = synth.uniform((B, 2, H, W), 0, 1, 6)
    sat = synth.heatmap((B, C, H, W), 41)
    sat[:, ::7, 20:90, 30:100] = 0.99                 # 4900 tied cells in every 7th plane
    sparse = np.zeros((B, C, H, W), np.float32)
    for i in range(30):
        sparse[i % B, (11 * i) % C, (17 * i) % H, (29 * i) % W] = 0.2 + 0.02 * i
    hot = synth.heatmap((B, C, H, W), 42) * 1e-3
    hot[:, 5] = synth.heatmap((B, H, W), 43)
    for heat in (sat, sparse, hot.astype(np.float32)):
        ref, ref_inds = cref.ctdet_decode(heat, wh, reg, K=K, return_inds=True)
        for flags in (0, 16384):
            d, i = ctdet_decode(_gpu(heat, dev), _gpu(wh, dev), _gpu(reg, dev), K=K, return_inds=True,
                                _debug_flags=flags)
            assert np.array_equal(i.cpu().numpy(), ref_inds)
            assert np.array_equal(d.cpu().numpy().view(np.uint32), ref.view(np.uint32))


def test_topk_plain_function_takes_the_one_launch_form(dev):
    """_topk without the _nms in front on negative and positive values (every cell ranks), through
    cn_topk_f32's one-launch form, vs the C oracle."""
    from centernet_amd.decode import _topk
    B, C, H, W, K = 2, 40, 128, 128, 50
    scores = synth.normal((B, C, H, W), 1.0, 23)
    ts, ti, tc, ty, tx = cref.topk(scores, K)
    s, i, c, ys, xs = _topk(_gpu(scores, dev), K=K, nms=False)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), ts.view(np.uint32))
    assert np.array_equal(i.cpu().numpy(), ti) and np.array_equal(c.cpu().numpy(), tc)
