"""The reference's public API on the HIP path: opts().init -> detector_factory[task](opt)
-> run(img) (README.md:101-116) and the prefetch-dict branch of test.py:35-42,70."""
import numpy as np
import pytest
import torch

from centernet_amd import synth
from oracle import net_oracle, post_oracle, pre_oracle

pytestmark = pytest.mark.gpu


def _detector(arch, extra=()):
    from centernet_amd.detectors import detector_factory
    from centernet_amd.opts import opts
    opt = opts().init(["ctdet", "--arch", arch] + list(extra))
    det = detector_factory[opt.task](opt)
    synth.fill_state_dict_(det.model, 317)
    return det, opt


def test_run_on_image_matches_oracle_pipeline(dev):
    det, opt = _detector("resdcn_18")
    rs = np.random.RandomState(0)
    image = rs.randint(0, 256, (512, 512, 3)).astype(np.uint8)       # BGR uint8, like cv2.imread
    ret = det.run(image)
    assert set(ret) == {"results", "tot", "load", "pre", "net", "dec", "post", "merge"}
    res = ret["results"]
    assert sorted(res) == list(range(1, 81))
    assert all(v.dtype == np.float32 and v.ndim == 2 and v.shape[1] == 5 for v in res.values())
    assert sum(len(v) for v in res.values()) == 100
    # oracle: same pre-processed tensor through the CPU restatement of the whole path
    images, meta = det.pre_process(image, 1.0)
    _, dets = net_oracle.ctdet_process("resdcn_18", det.model.state_dict(), images, list(opt.heads), K=opt.K)
    ref = post_oracle.ctdet_results(dets, meta, opt.num_classes)
    n_same = 0
    for j in range(1, 81):
        if len(ref[j]) == len(res[j]) and len(ref[j]):
            # pair boxes by position (two boxes of a class can have near-equal scores)
            a = res[j][np.lexsort((res[j][:, 1], np.round(res[j][:, 0])))]
            b = ref[j][np.lexsort((ref[j][:, 1], np.round(ref[j][:, 0])))]
            assert np.abs(a[:, 4] - b[:, 4]).max() < 1e-4
            assert np.abs(a[:, :4] - b[:, :4]).max() < 2e-3       # image pixels (x4 the grid)
            n_same += len(a)
    assert n_same >= 95


def _oracle_inputs(image, scale, opt, flip_test=False):
    """base_detector.py:37-65 through the scalar oracle (bit-identical to the product's host
    and device pre-process, tests/test_oracle_pre.py / test_gpu_pre.py)."""
    images, meta = pre_oracle.pre_process(image, scale, opt.mean, opt.std, fix_res=opt.fix_res,
                                          input_h=opt.input_h, input_w=opt.input_w, pad=opt.pad,
                                          flip_test=flip_test, down_ratio=opt.down_ratio)
    return torch.from_numpy(images), meta


def _compare_class_rows(res, ref, n_min, score_tol=1e-4, box_tol=2e-3):
    """Per class: same number of rows, rows paired by position, scores / boxes within
    tolerance; returns how many rows were compared (classes whose row count differs -- a
    detection at the top-100 threshold that flipped -- are skipped and bounded by n_min)."""
    n_same = 0
    for j in sorted(ref):
        if len(ref[j]) == len(res[j]) and len(ref[j]):
            a = res[j][np.lexsort((res[j][:, 1], np.round(res[j][:, 0])))]
            b = ref[j][np.lexsort((ref[j][:, 1], np.round(ref[j][:, 0])))]
            assert np.abs(a[:, 4] - b[:, 4]).max() < score_tol, j
            assert np.abs(a[:, :4] - b[:, :4]).max() < box_tol, j
            n_same += len(a)
    assert n_same >= n_min, n_same
    return n_same


def test_flip_test_matches_oracle_pipeline(dev):
    """--flip_test (detectors/ctdet.py:34-37): the mirrored frame is averaged in after the
    sigmoid -- against the oracle's flip pipeline, not against the product itself."""
    det, opt = _detector("resdcn_18", ["--flip_test"])
    image = np.random.RandomState(1).randint(0, 256, (512, 512, 3)).astype(np.uint8)
    res = det.run(image)["results"]
    images, meta = _oracle_inputs(image, 1.0, opt, flip_test=True)
    assert tuple(images.shape) == (2, 3, 512, 512)
    _, dets = net_oracle.ctdet_process("resdcn_18", det.model.state_dict(), images, list(opt.heads),
                                       K=opt.K, flip_test=True)
    ref = post_oracle.ctdet_results(dets, meta, opt.num_classes)
    assert sum(len(v) for v in res.values()) == 100
    _compare_class_rows(res, ref, 95)


def test_multi_scale_soft_nms_matches_oracle_pipeline(dev):
    """--test_scales 1,0.75 --nms (detectors/ctdet.py:58-73, base_detector.py:99-127): per-scale
    pre-process / network / decode / post-process, merge, soft-NMS (pinned to the reference's
    cython, tests/test_oracle_ref.py), top-100 threshold -- every stage from the oracle."""
    det, opt = _detector("resdcn_18", ["--test_scales", "1,0.75", "--nms", "--keep_res"])
    image = np.random.RandomState(3).randint(0, 256, (200, 264, 3)).astype(np.uint8)
    res = det.run(image)["results"]
    per_scale = []
    for scale in opt.test_scales:
        images, meta = _oracle_inputs(image, scale, opt)
        _, dets = net_oracle.ctdet_process("resdcn_18", det.model.state_dict(), images,
                                           list(opt.heads), K=opt.K)
        per_scale.append(post_oracle.ctdet_post_process_scale(dets, meta, opt.num_classes, scale))
    ref = post_oracle.ctdet_merge_outputs(per_scale, opt.num_classes, len(opt.test_scales), nms=True)
    n_ref = sum(len(v) for v in ref.values())
    n_res = sum(len(v) for v in res.values())
    assert 100 <= n_ref <= 200 and abs(n_res - n_ref) <= 2
    # soft-NMS decays scores by overlap, so score agreement also checks the box agreement
    _compare_class_rows(res, ref, int(0.9 * n_ref), score_tol=2e-4, box_tol=4e-3)


def test_prefetch_dict_branch_and_flip_test(dev):
    det, opt = _detector("resdcn_18", ["--flip_test"])
    image = np.random.RandomState(1).randint(0, 256, (512, 512, 3)).astype(np.uint8)
    images, meta = det.pre_process(image, 1.0)
    assert tuple(images.shape) == (2, 3, 512, 512)               # flip_test doubles the batch
    pre = {"images": {1.0: images[None]}, "image": torch.from_numpy(image)[None],
           "meta": {1.0: {k: torch.from_numpy(np.asarray(v))[None] for k, v in meta.items()}}}
    r1 = det.run(pre)["results"]
    r2 = det.run(image)["results"]
    for j in range(1, 81):
        assert np.array_equal(r1[j], r2[j])


def test_run_batch(dev):
    det, opt = _detector("resdcn_18")
    x = synth.images(4, 512, 512, seed=2).to(dev)
    d = det.run_batch(x)
    assert tuple(d.shape) == (4, 100, 6)
    one = det.run_batch(x[1:2].contiguous())
    # kernels pick tile / split-K shapes per batch size, so a different batch size may round
    # differently: same detections within fp32 noise (scores 1e-5), not bit-identical
    assert torch.equal(one[0, :, 5], d[1, :, 5])
    assert float((one[0] - d[1]).abs().max()) < 1e-3 and float((one[0, :, 4] - d[1, :, 4]).abs().max()) < 1e-5
    assert torch.equal(det.run_batch(x), d)            # same batch: bit-identical replay


def test_multi_pose_detector_matches_oracle(dev):
    """multi_pose task on DLA-34 (BASELINE configs[3]): run(img) -> {1: [[39 floats], ...]};
    compared with the CPU restatement of network + multi_pose_decode + post-process."""
    from centernet_amd.detectors import detector_factory
    from centernet_amd.opts import opts
    from oracle import cref
    opt = opts().init(["multi_pose", "--arch", "dla_34"])
    det = detector_factory[opt.task](opt)
    synth.fill_state_dict_(det.model, 317)
    image = np.random.RandomState(5).randint(0, 256, (512, 512, 3)).astype(np.uint8)
    ret = det.run(image)
    res = ret["results"]
    assert list(res.keys()) == [1] and np.array(res[1]).shape == (100, 39)
    got = np.array(res[1], np.float32)
    images, meta = det.pre_process(image, 1.0)
    out = net_oracle.forward("dla_34", det.model.state_dict(), images, list(opt.heads))
    hm = out["hm"].sigmoid_().numpy()
    hm_hp = out["hm_hp"].sigmoid_().numpy()
    dets = cref.multi_pose_decode(hm, out["wh"].numpy(), out["hps"].numpy(), out["reg"].numpy(),
                                  hm_hp, out["hp_offset"].numpy(), K=opt.K)
    ref = np.array(post_oracle.multi_pose_results(dets, meta)[1], np.float32)
    assert np.abs(got[:, 4] - ref[:, 4]).max() < 1e-4            # scores
    from oracle.parity import compare_topk
    r = compare_topk(got[None], ref[None], box_tol=1e-5)         # 1e-5 x 512 px = 5e-3 px
    print("multi_pose detector: paired %.3f, same rank %.3f" % (r["paired"], r["in_place"]))
    assert r["paired"] >= 0.99
    gap = np.minimum(np.abs(np.diff(ref[:, 4], prepend=np.inf)), np.abs(np.diff(ref[:, 4], append=-np.inf)))
    safe = gap > 2e-6
    assert np.abs(got[safe, :4] - ref[safe, :4]).max() < 5e-3    # boxes, image pixels
    # keypoints: regression branch within 5e-3 px; the heat-map-snapped ones are discrete
    # choices, so allow a few to differ where the reject rule sits on its threshold
    kd = np.abs(got[safe, 5:] - ref[safe, 5:])
    assert (kd < 5e-3).mean() > 0.97


def test_multi_pose_flip_test_matches_oracle_pipeline(dev):
    """multi_pose --flip_test (detectors/multi_pose.py:44-55): flip_lr / flip_lr_off on the
    device (index permutation) vs the reference's NumPy round trip restated in the oracle."""
    from centernet_amd.detectors import detector_factory
    from centernet_amd.opts import opts
    opt = opts().init(["multi_pose", "--arch", "dla_34", "--flip_test", "--input_res", "256"])
    det = detector_factory[opt.task](opt)
    synth.fill_state_dict_(det.model, 317)
    image = np.random.RandomState(7).randint(0, 256, (256, 256, 3)).astype(np.uint8)
    got = np.array(det.run(image)["results"][1], np.float32)
    images, meta = _oracle_inputs(image, 1.0, opt, flip_test=True)
    _, dets = net_oracle.multi_pose_process("dla_34", det.model.state_dict(), images,
                                            list(opt.heads), K=opt.K, flip_test=True,
                                            flip_idx=opt.flip_idx)
    ref = np.array(post_oracle.multi_pose_results(dets, meta)[1], np.float32)
    assert got.shape == ref.shape == (100, 39)
    assert np.abs(got[:, 4] - ref[:, 4]).max() < 1e-4
    gap = np.minimum(np.abs(np.diff(ref[:, 4], prepend=np.inf)), np.abs(np.diff(ref[:, 4], append=-np.inf)))
    safe = gap > 2e-6
    from oracle.parity import compare_topk
    r = compare_topk(got[None], ref[None], box_tol=2e-5)
    assert r["paired"] >= 0.99, r
    assert np.abs(got[safe, :4] - ref[safe, :4]).max() < 5e-3
    assert (np.abs(got[safe, 5:] - ref[safe, 5:]) < 5e-3).mean() > 0.97


def test_multi_pose_run_frames_equals_run(dev):
    """The batched surface for the pose task: run_frames == run() per frame."""
    import contextlib, sys
    from centernet_amd.opts import opts
    from centernet_amd.detectors.detector_factory import detector_factory
    with contextlib.redirect_stdout(sys.stderr):
        opt = opts().init(["multi_pose", "--arch", "dla_34", "--input_h", "128", "--input_w", "128"])
        det = detector_factory[opt.task](opt)
    synth.fill_state_dict_(det.model, 317)
    det.model.invalidate_plans()
    rng = np.random.RandomState(6)
    frames = [rng.randint(0, 256, (96, 120, 3)).astype(np.uint8) for _ in range(3)]
    batched = det.run_frames(frames)
    assert len(batched) == 3
    for f, rb in zip(frames, batched):
        rs = det.run(f)['results']
        a, b = np.array(rb[1], np.float32), np.array(rs[1], np.float32)
        assert a.shape == b.shape == (opt.K, 39)
        assert np.abs(a[:, 4] - b[:, 4]).max() < 1e-4          # scores
        assert np.abs(a - b).max() < 5e-3                      # pixels


def test_run_frames_stream_equals_run_frames_and_the_host_tail(dev):
    """run_frames_stream (pinned staging, asynchronous upload on a copy stream, batched device
    pre-process, device tail cn_ctdet_post_process_f32, no host synchronisation between batches)
    returns, batch by batch and in order, exactly what run_frames returns for each batch alone;
    and the device tail is bit-identical to the host tail (ctdet_results_batch: the reference's
    float64 affine + float32 rounding + per-class split, utils/post_process.py:83-100) on the same
    raw detections -- including test scale 0.5 (the resize in front of the warp, boxes / scale)."""
    import contextlib, sys
    from centernet_amd.opts import opts
    from centernet_amd.detectors.detector_factory import detector_factory
    from centernet_amd.post_process import ctdet_results_batch
    for scales in ("1", "0.5"):
        with contextlib.redirect_stdout(sys.stderr):
            opt = opts().init(["ctdet", "--arch", "resdcn_18", "--input_h", "128", "--input_w", "128",
                               "--test_scales", scales])
            det = detector_factory[opt.task](opt)
        synth.fill_state_dict_(det.model, 317)
        det.model.invalidate_plans()
        rng = np.random.RandomState(7)
        batches = [[rng.randint(0, 256, (100, 140, 3)).astype(np.uint8) for _ in range(4)] for _ in range(5)]
        alone = [det.run_frames(b) for b in batches]
        streamed = list(det.run_frames_stream(iter(batches), depth=2))
        assert len(streamed) == len(batches)
        for ra, rs in zip(alone, streamed):
            assert len(ra) == len(rs) == 4
            for a, b in zip(ra, rs):
                for j in range(1, 81):
                    assert a[j].dtype == np.float32 and np.array_equal(a[j], b[j]), j
        # device tail vs host tail on the same raw detections
        pipe = det._pipe_for(batches[0], 1)
        assert pipe.tail is not None
        pipe.submit(0, batches[0])
        got = pipe.collect(0, batches[0])
        raw = det.run_batch(pipe.batch).detach().cpu().numpy()
        want = ctdet_results_batch(raw, [pipe.meta] * 4, det.opt.num_classes, pipe.scale, det.max_per_image)
        for a, b in zip(got, want):
            for j in range(1, 81):
                assert a[j].shape == b[j].shape and np.array_equal(a[j].view(np.int32), b[j].view(np.int32)), j


def test_run_batch_without_a_range_look_is_reported(dev):
    """run_batch never synchronises, so it cannot look at the f32s range words itself: a caller
    that keeps calling it without range_ok() is warned (once per limit); a look resets the count."""
    import contextlib, sys, warnings
    from centernet_amd.opts import opts
    from centernet_amd.detectors.detector_factory import detector_factory
    with contextlib.redirect_stdout(sys.stderr):
        opt = opts().init(["ctdet", "--arch", "res_18", "--input_h", "128", "--input_w", "128"])
        det = detector_factory[opt.task](opt)
    synth.fill_state_dict_(det.model, 317)
    det.model.invalidate_plans()
    det.UNCHECKED_LIMIT = 3
    x = synth.images(2, 128, 128, seed=1).to(dev)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        det.run_batch(x); det.run_batch(x)
        assert not [w for w in rec if "range_ok" in str(w.message)]
        assert det.range_ok()
        det.run_batch(x); det.run_batch(x)
        assert not [w for w in rec if "range_ok" in str(w.message)]
        det.run_batch(x)
        assert len([w for w in rec if "range_ok" in str(w.message)]) == 1


def test_debug_levels_run_like_debug_0(dev):
    """--debug 1 / 2 (base_detector.py:85-87,127-141: Debugger windows) are outside the hot path: the
    detector runs as with --debug 0, returns the same results, and says once that nothing is drawn."""
    import contextlib, sys, warnings
    from centernet_amd.opts import opts
    from centernet_amd.detectors.detector_factory import detector_factory
    rng = np.random.RandomState(3)
    frame = rng.randint(0, 256, (100, 140, 3)).astype(np.uint8)
    res = {}
    for dbg in (0, 2):
        with contextlib.redirect_stdout(sys.stderr):
            opt = opts().init(["ctdet", "--arch", "res_18", "--input_h", "128", "--input_w", "128",
                               "--debug", str(dbg)])
            det = detector_factory[opt.task](opt)
        synth.fill_state_dict_(det.model, 317)
        det.model.invalidate_plans()
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            res[dbg] = det.run(frame)['results']
            det.run(frame)
        assert len([w for w in rec if "visual debugging" in str(w.message)]) == (1 if dbg else 0)
    for j in range(1, 81):
        assert np.array_equal(res[0][j], res[2][j])


def test_flip_average_kernel_equals_the_reference_formulas(dev):
    """cn_flip_average_f32 against (a + flip(b)) / 2 with flip = flip_tensor / flip_lr / flip_lr_off of
    models/utils.py:28-50 (written out with torch ops on the CPU): bit-identical without the logistic,
    within 1e-6 with it (in place on both images, as hm.sigmoid_() does)."""
    from centernet_amd.utils import flip_average
    g = torch.Generator().manual_seed(3)
    flip_idx = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
    perm = list(range(17))
    for a, b in flip_idx:
        perm[a], perm[b] = perm[b], perm[a]
    hm = torch.randn((2, 80, 24, 40), generator=g)
    want = (hm[0:1] + torch.flip(hm[1:2], [3])) / 2
    assert torch.equal(flip_average(hm.to(dev)).cpu(), want)
    hp = torch.randn((2, 17, 24, 40), generator=g)
    want = (hp[0:1] + torch.flip(hp[1:2], [3])[:, perm]) / 2
    assert torch.equal(flip_average(hp.to(dev), flip_idx).cpu(), want)
    off = torch.randn((2, 34, 24, 40), generator=g)
    f = torch.flip(off[1:2], [3]).reshape(1, 17, 2, 24, 40).clone()
    f[:, :, 0] *= -1
    want = (off[0:1] + f[:, perm].reshape(1, 34, 24, 40)) / 2
    assert torch.equal(flip_average(off.to(dev), flip_idx, offsets=True).cpu(), want)
    d = hm.to(dev)
    got = flip_average(d, sigmoid=True).cpu()
    s = hm.sigmoid()
    assert float((got - (s[0:1] + torch.flip(s[1:2], [3])) / 2).abs().max()) < 1e-6
    assert float((d.cpu() - s).abs().max()) < 1e-6            # both images now hold scores


def test_run_frames_equals_run(dev):
    """run_frames (batched, device pre-process) == run() per frame (same kernels per image up to
    the batch-size dependent split-K summation order)."""
    import contextlib, sys
    from centernet_amd.opts import opts
    from centernet_amd.detectors.detector_factory import detector_factory
    with contextlib.redirect_stdout(sys.stderr):
        opt = opts().init(["ctdet", "--arch", "resdcn_18", "--input_h", "128", "--input_w", "128"])
        det = detector_factory[opt.task](opt)
    synth.fill_state_dict_(det.model, 317)
    det.model.invalidate_plans()
    rng = np.random.RandomState(5)
    frames = [rng.randint(0, 256, (100, 140, 3)).astype(np.uint8) for _ in range(3)]
    batched = det.run_frames(frames)
    for f, rb in zip(frames, batched):
        rs = det.run(f)['results']
        for j in range(1, 81):
            assert rb[j].shape == rs[j].shape
            if len(rb[j]):
                assert np.abs(rb[j] - rs[j]).max() < 2e-3
