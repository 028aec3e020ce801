"""The ddd and exdet tasks end to end on the HIP path -- opts().init -> detector_factory[task](opt) ->
run(image[, calib]) (src/lib/detectors/ddd.py, exdet.py; test.py:37-39,105-106) -- against the oracle's
pipeline: the SAME pre-processed tensor through the CPU restatement of the network, the decode, the
post-process and the merge (every stage pinned to the reference on the CPU side:
tests/test_tasks_host.py, test_oracle_ddd.py, test_oracle_exct.py, test_oracle_net.py)."""
import json
import os

import numpy as np
import pytest
import torch

from centernet_amd import synth
from oracle import net_oracle, post_oracle
from oracle.parity import match_rows

pytestmark = pytest.mark.gpu

KITTI_CALIB = [[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791],
               [0.0, 0.0, 1.0, 0.002745884]]


def _note(name, **kw):
    """measured agreement, kept next to the other parity fractions (gpurun_out/parity_fractions.jsonl)"""
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/parity_fractions.jsonl", "a") as f:
            f.write(json.dumps(dict(test=name, **kw)) + "\n")
    except OSError:
        pass


def _detector(args):
    from centernet_amd.detectors import detector_factory
    from centernet_amd.opts import opts
    opt = opts().init(args)
    det = detector_factory[opt.task](opt)
    synth.fill_state_dict_(det.model, 317)
    return det, opt


def _favour_one_class(model, cls=17, by=3.0):
    """the last layer of the five exdet maps (the only 80-wide biases of those heads) gets one strong class"""
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.split(".")[0] in ("hm_t", "hm_l", "hm_b", "hm_r", "hm_c") and k.endswith("bias") and v.numel() == 80:
                v[cls] += by


def _paired_fraction(got, ref, cols, atol, rtol, window):
    """share of the oracle's rows that have a partner among the produced rows at most ``window`` ranks away
    with every column of ``cols`` within atol + rtol * |ref|"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if len(ref) == 0:
        return 1.0 if len(got) == 0 else 0.0
    used = np.zeros(len(got), bool)
    hit = 0
    for r in range(len(ref)):
        lo, hi = max(0, r - window), min(len(got), r + window + 1)
        for j in sorted(range(lo, hi), key=lambda j: abs(j - r)):
            if not used[j] and np.all(np.abs(got[j, cols] - ref[r, cols]) <= atol + rtol * np.abs(ref[r, cols])):
                used[j] = True
                hit += 1
                break
    return hit / float(len(ref))


def test_ddd_detector_matches_the_oracle_pipeline(dev):
    """KITTI-sized frame through dla_34 at the task's default 384 x 1280 input (a 96 x 320 output grid):
    raw rows of process(), the lifted rows of post_process(), and run()'s results with both calibrations."""
    det, opt = _detector(["ddd"])
    assert opt.arch == "dla_34" and (opt.input_h, opt.input_w) == (384, 1280) and opt.K == 100
    image = np.random.RandomState(11).randint(0, 256, (375, 1242, 3)).astype(np.uint8)
    images, meta = det.pre_process(image, 1.0, KITTI_CALIB)
    sd = det.model.state_dict()
    _, ref_dets = net_oracle.ddd_process("dla_34", sd, images, list(opt.heads), K=opt.K)
    # ---- raw rows: [x, y, score, rot 8, depth, dim 3, wh 2, class]
    output, dets = det.process(images.to(dev))
    got = dets.cpu().numpy()
    assert got.shape == ref_dets.shape == (1, 100, 18)
    m = match_rows(got[0], ref_dets[0], [0, 1, 2, 17], np.array([2e-3, 2e-3, 1e-4, 0.0]), window=8)
    paired = float((m >= 0).mean())
    ok = m >= 0
    err = np.abs(got[0][ok] - ref_dets[0][m[ok]])
    rel = err / (1e-3 + np.abs(ref_dets[0][m[ok]]))
    _note("ddd_raw", paired=paired, in_place=float((m == np.arange(100)).mean()), score_err=float(err[:, 2].max()),
          col_rel_err=[float(v) for v in rel.max(axis=0)])
    assert paired >= 0.97, paired
    assert err[:, 2].max() < 1e-4                                     # scores
    assert np.array_equal(got[0][ok][:, 17], ref_dets[0][m[ok]][:, 17])   # classes identical on paired rows
    # gathered heads (rot, depth, dim, wh): the network bar of 1e-4 absolute, 2e-3 relative on top for
    # the depth, which is 1 / sigmoid of its map
    ref_g = np.abs(ref_dets[0][m[ok]][:, 3:17])
    assert (err[:, 3:17] <= 2e-4 + 2e-3 * ref_g).mean() > 0.995
    # the returned maps are in the reference's transformed state (ddd.py:59-60)
    assert float(output["hm"].min()) >= 0.0 and float(output["hm"].max()) <= 1.0 and float(output["dep"].min()) > -1.0
    # ---- lifted rows
    res = det.post_process(dets, meta)
    ref = post_oracle.ddd_results(ref_dets, meta, opt.num_classes, opt.output_w, opt.output_h)
    assert det.this_calib is meta["calib"]
    n_rows = 0
    for j in (1, 2, 3):
        assert res[j].dtype == np.float32 and (res[j].shape == (0,) or res[j].shape[1] == 13)
        if len(ref[j]) == 0:
            assert len(res[j]) <= 1
            continue
        # [alpha, box 4, dims 3, location 3, rotation_y, score]; angles are compared modulo the bin choice
        # by leaving them to the dedicated share below
        frac = _paired_fraction(res[j], ref[j], [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12], 2e-2, 2e-3, window=8)
        ang = _paired_fraction(res[j], ref[j], [0, 11, 12], 2e-3, 1e-3, window=8)
        _note("ddd_lifted", cls=j, rows=len(ref[j]), paired=frac, angles=ang)
        assert abs(len(res[j]) - len(ref[j])) <= 1 and frac >= 0.95 and ang >= 0.9, (j, frac, ang)
        n_rows += len(ref[j])
    assert n_rows == 100
    # ---- the public entry: run(image, calib) and run(image) (the detector's default matrix)
    # --peak_thresh inside the score range of these synthetic weights, in the middle of the widest gap
    # between two neighbouring oracle scores of the central ranks: a real cut that no rounding can move
    sc = np.sort(ref_dets[0, :, 2])[::-1]
    k = 30 + int(np.argmax(sc[30:70] - sc[31:71]))
    assert sc[k] - sc[k + 1] > 1e-5
    opt.peak_thresh = float((sc[k] + sc[k + 1]) / 2)
    ret = det.run(image, KITTI_CALIB)
    assert set(ret) == {"results", "tot", "load", "pre", "net", "dec", "post", "merge"}
    results = ret["results"]
    merged = post_oracle.ddd_merge_outputs([ref], opt.num_classes, opt.peak_thresh)
    kept = sum(len(merged[j]) for j in (1, 2, 3))
    assert 0 < kept < 100, kept            # the threshold was chosen inside the score range: a real cut
    for j in (1, 2, 3):
        assert abs(len(results[j]) - len(merged[j])) <= 1
        if len(merged[j]):
            assert np.all(results[j][:, -1] > opt.peak_thresh)
            assert _paired_fraction(results[j], merged[j], [1, 2, 3, 4, 8, 9, 10, 12], 2e-2, 2e-3, window=8) >= 0.95
    default = det.run(image)["results"]
    for j in (1, 2, 3):
        if len(results[j]) and len(default[j]) == len(results[j]):
            assert np.array_equal(default[j][:, [0, 1, 2, 3, 4, 5, 6, 7, 12]], results[j][:, [0, 1, 2, 3, 4, 5, 6, 7, 12]])
            assert not np.array_equal(default[j][:, 8:11], results[j][:, 8:11])     # another camera: other locations


def test_ddd_run_batch_equals_process(dev):
    """New surface: run_batch (sigmoid fused into the decode) against process() on the same tensor."""
    det, opt = _detector(["ddd", "--input_h", "128", "--input_w", "384", "--K", "40"])
    x = synth.images(2, 128, 384, seed=4).to(dev)
    raw = det.run_batch(x).cpu().numpy()
    assert raw.shape == (2, 40, 18) and det.range_ok()
    for b in range(2):
        _, one = det.process(x[b:b + 1].contiguous())
        one = one.cpu().numpy()[0]
        m = match_rows(raw[b], one, [0, 1, 2, 17], np.array([2e-3, 2e-3, 5e-5, 0.0]), window=6)
        assert (m >= 0).mean() >= 0.95


def test_exdet_detector_matches_the_oracle_pipeline(dev):
    """ExtremeNet-style task on the hourglass with --flip_test (how the task is run: exdet.py:87-91), nine
    heads = two fused head launches (6 + 3); thresholds at 0 so that the synthetic weights leave valid groupings."""
    det, opt = _detector(["exdet", "--arch", "hourglass", "--flip_test", "--input_res", "256", "--K", "40",
                          "--scores_thresh", "0", "--center_thresh", "0"])
    assert list(opt.heads) == ["hm_t", "hm_l", "hm_b", "hm_r", "hm_c", "reg_t", "reg_l", "reg_b", "reg_r"]
    # random heads never agree on a class (every grouping would carry the class rejection, score < 0):
    # one class is favoured in the last layer of the five maps, as a trained net's dominant object would be
    _favour_one_class(det.model)
    image = np.random.RandomState(12).randint(0, 256, (256, 256, 3)).astype(np.uint8)
    images, meta = det.pre_process(image, 1.0)
    assert tuple(images.shape) == (2, 3, 256, 256)
    sd = det.model.state_dict()
    out_ref, ref_dets = net_oracle.exdet_process("hourglass", sd, images, list(opt.heads), K=opt.K,
                                                 scores_thresh=0.0, center_thresh=0.0)
    output, dets = det.process(images.to(dev))
    got = dets.cpu().numpy()
    assert got.shape == ref_dets.shape == (2, 1000, 14)
    # the maps themselves (post-sigmoid in place, as the reference leaves them)
    for n in ("hm_t", "hm_c", "reg_l", "reg_r"):
        r = out_ref[n].numpy()
        assert np.abs(output[n].cpu().numpy() - r).max() < 1e-4 * max(1.0, float(np.abs(r).max())), n
    # rows: [x1, y1, x2, y2, score, t(x, y), l, b, r, class]; scores are sums of five map values, so ranks
    # within ~1e-6 of each other trade places: pair within a window instead of by rank
    tol = np.array([2e-3] * 4 + [1e-4] + [2e-3] * 8 + [0.0])
    fr = []
    for b in range(2):
        top = 300
        m = match_rows(got[b][:top + 20], ref_dets[b][:top + 20], list(range(14)), tol, window=20)
        fr.append(float((m[:top] >= 0).mean()))
    valid = int((ref_dets[..., 4] > 0).sum())
    _note("exdet_raw", paired=fr, valid_rows=valid)
    assert min(fr) >= 0.95, fr
    assert valid >= 1000, valid
    # ---- run(): post_process + merge_outputs against the oracle's
    res = det.run(image)["results"]
    rows = post_oracle.exdet_post_process(ref_dets, meta, 1.0)
    ref = post_oracle.exdet_merge_outputs([rows], opt.num_classes)
    assert sorted(res) == list(range(1, 81))
    n_ref = sum(len(v) for v in ref.values())
    n_res = sum(len(v) for v in res.values())
    assert n_ref > 0 and abs(n_res - n_ref) <= max(2, n_ref // 20), (n_res, n_ref)
    same = 0
    for j in range(1, 81):
        assert res[j].dtype == np.float32 and res[j].ndim == 2 and res[j].shape[1] == 5
        if len(ref[j]) and len(res[j]):
            # soft-NMS leaves the rows in its own greedy order: pair every oracle row with its nearest row
            d = np.abs(res[j][:, None, :].astype(np.float64) - ref[j][None, :, :]).max(axis=2)
            same += int((d.min(axis=0) < 5e-3).sum())
    _note("exdet_results", rows=n_ref, same=same)
    assert same >= 0.9 * n_ref, (same, n_ref)
    # the task's guard rails
    from centernet_amd.opts import opts
    from centernet_amd.detectors import detector_factory
    with pytest.raises(ValueError):
        detector_factory["exdet"](opts().init(["exdet", "--arch", "hourglass", "--K", "100"]))      # an explicit --K > 64
    assert opts().init(["exdet", "--arch", "hourglass"]).K == 40     # no --K: ExtremeNet's own setting, not the 100 of the other tasks


# ------------------------------------------------------------------------------------------------
# --agnostic_ex: agnex_ct_decode (models/decode.py:121-271) through cn_agnex_ct_decode_f32
# ------------------------------------------------------------------------------------------------
from test_oracle_agnex import GEN as AGEN, GOLD as AGOLD      # noqa: E402
from test_oracle_exct import assert_same                      # noqa: E402


def _t(a, dev):
    return None if a is None else torch.from_numpy(a).to(dev)


@pytest.mark.parametrize("name", sorted(AGEN.AGNEX_CASES))
def test_agnex_ct_decode_matches_reference_golden(dev, name):
    from centernet_amd import decode as D
    from oracle import cref
    heats, regs, K, num_dets = AGEN.agnex_inputs(name)
    dets = D.agnex_ct_decode(*[_t(h, dev) for h in heats], *[_t(r, dev) for r in regs], K=K,
                             num_dets=num_dets).cpu().numpy()
    assert_same(dets, AGOLD[name + "/dets"])
    ref = cref.agnex_ct_decode(*heats, *regs, K=K, num_dets=num_dets)
    assert np.array_equal(dets.view(np.uint32), ref.view(np.uint32))     # same tie rule: fully bit-exact
    if name == "agnex_small":
        with pytest.raises(RuntimeError):        # edge maps with classes belong to exct_decode
            D.agnex_ct_decode(*[_t(np.repeat(h, 2, axis=1) if i < 4 else h, dev) for i, h in enumerate(heats)], K=K,
                              num_dets=num_dets)
        # with the edge aggregation in front (the shared cn_exct_aggregate_f32 + clamp path)
        got = D.agnex_ct_decode(*[_t(h, dev) for h in heats], *[_t(r, dev) for r in regs], K=K,
                                num_dets=num_dets, aggr_weight=0.1).cpu().numpy()
        ref = cref.agnex_ct_decode(*heats, *regs, K=K, num_dets=num_dets, aggr_weight=0.1)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_exdet_agnostic_detector_matches_the_oracle_pipeline(dev):
    """--agnostic_ex end to end: one extreme-point map per edge, classes from the centre map's arg-max."""
    det, opt = _detector(["exdet", "--arch", "hourglass", "--agnostic_ex", "--flip_test", "--input_res", "256",
                          "--K", "40", "--scores_thresh", "0", "--center_thresh", "0"])
    assert [opt.heads[n] for n in ("hm_t", "hm_l", "hm_b", "hm_r", "hm_c")] == [1, 1, 1, 1, 80]
    image = np.random.RandomState(13).randint(0, 256, (256, 256, 3)).astype(np.uint8)
    images, meta = det.pre_process(image, 1.0)
    _, ref_dets = net_oracle.exdet_process("hourglass", det.model.state_dict(), images, list(opt.heads), K=opt.K,
                                           scores_thresh=0.0, center_thresh=0.0, agnostic=True)
    _, dets = det.process(images.to(dev))
    got = dets.cpu().numpy()
    assert got.shape == ref_dets.shape == (2, 1000, 14)
    tol = np.array([2e-3] * 4 + [1e-4] + [2e-3] * 8 + [0.0])
    fr = []
    for b in range(2):
        m = match_rows(got[b][:320], ref_dets[b][:320], list(range(14)), tol, window=20)
        fr.append(float((m[:300] >= 0).mean()))
    valid = int((ref_dets[..., 4] > 0).sum())
    classes = len(set(ref_dets[..., 13].reshape(-1).tolist()))
    _note("exdet_agnostic_raw", paired=fr, valid_rows=valid, classes=classes)
    assert min(fr) >= 0.95 and valid >= 1000 and classes >= 1, (fr, valid, classes)
    res = det.run(image)["results"]
    ref = post_oracle.exdet_merge_outputs([post_oracle.exdet_post_process(ref_dets, meta, 1.0)], opt.num_classes)
    n_ref = sum(len(v) for v in ref.values())
    same = 0
    for j in range(1, 81):
        if len(ref[j]) and len(res[j]):
            d = np.abs(res[j][:, None, :].astype(np.float64) - ref[j][None, :, :]).max(axis=2)
            same += int((d.min(axis=0) < 5e-3).sum())
    _note("exdet_agnostic_results", rows=n_ref, same=same)
    assert n_ref > 0 and same >= 0.9 * n_ref, (same, n_ref)


@pytest.mark.parametrize("arch", ["resdcn_18", "dla_34"])
def test_exdet_on_the_other_backbones(dev, arch):
    """The zoo also carries exdet on DLA-34; head_conv = 64 (resdcn_18) puts the nine heads on the persistent
    kernel's fused-heads form, 256 (dla_34) on the wide form -- both as two launches of six and three heads."""
    det, opt = _detector(["exdet", "--arch", arch, "--flip_test", "--input_res", "256", "--K", "40",
                          "--scores_thresh", "0", "--center_thresh", "0"])
    _favour_one_class(det.model)
    image = np.random.RandomState(14).randint(0, 256, (256, 256, 3)).astype(np.uint8)
    images, meta = det.pre_process(image, 1.0)
    out_ref, ref_dets = net_oracle.exdet_process(arch, det.model.state_dict(), images, list(opt.heads), K=opt.K,
                                                 scores_thresh=0.0, center_thresh=0.0)
    output, dets = det.process(images.to(dev))
    for n in opt.heads:
        r = out_ref[n].numpy()
        assert np.abs(output[n].cpu().numpy() - r).max() < 1e-4 * max(1.0, float(np.abs(r).max())), n
    got = dets.cpu().numpy()
    tol = np.array([2e-3] * 4 + [1e-4] + [2e-3] * 8 + [0.0])
    fr = []
    for b in range(2):
        m = match_rows(got[b][:320], ref_dets[b][:320], list(range(14)), tol, window=20)
        fr.append(float((m[:300] >= 0).mean()))
    _note("exdet_" + arch, paired=fr, valid_rows=int((ref_dets[..., 4] > 0).sum()))
    assert min(fr) >= 0.95, fr
