"""The reference's public API on the HIP path: opts().init -> detector_factory[task](opt)
-> run(img) (README.md:101-116) and the prefetch-dict branch of test.py:35-42,70."""
import numpy as np
import pytest
import torch

from centernet_amd import synth
from oracle import net_oracle, post_oracle

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


def test_multi_scale_with_soft_nms(dev):
    """--test_scales 1,0.75 --nms: two passes at different resolutions, detections merged per
    class and decayed by the native soft-NMS (detectors/ctdet.py:58-73)."""
    det, opt = _detector("resdcn_18", ["--test_scales", "1,0.75", "--nms"])
    image = np.random.RandomState(3).randint(0, 256, (384, 512, 3)).astype(np.uint8)
    res = det.run(image)["results"]
    assert sorted(res) == list(range(1, 81))
    n = sum(len(v) for v in res.values())
    assert 100 <= n <= 200          # top-100 threshold keeps ties, at most 2x100 candidates
    allb = np.concatenate([v for v in res.values() if len(v)], 0)
    assert np.isfinite(allb).all() and (allb[:, 4] > 0).all() and (allb[:, 4] <= 1).all()


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
    from centernet_amd.post_process import multi_pose_post_process
    ref = np.array(multi_pose_post_process(dets.copy(), [meta["c"]], [meta["s"]], meta["out_height"],
                                           meta["out_width"])[0][1], np.float32)
    assert np.abs(got[:, 4] - ref[:, 4]).max() < 1e-4            # scores
    same = np.abs(got[:, 4] - ref[:, 4]) < 1e-6
    gap = np.minimum(np.abs(np.diff(ref[:, 4], prepend=np.inf)), np.abs(np.diff(ref[:, 4], append=-np.inf)))
    safe = gap > 2e-6
    assert safe.mean() > 0.8
    assert np.abs(got[safe, :4] - ref[safe, :4]).max() < 5e-3    # boxes, image pixels
    # keypoints: regression branch within 5e-3 px; the heat-map-snapped ones are discrete
    # choices, so allow a few to differ where the reject rule sits on its threshold
    kd = np.abs(got[safe, 5:] - ref[safe, 5:])
    assert (kd < 5e-3).mean() > 0.97


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
