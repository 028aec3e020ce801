"""Host side of the ddd and exdet tasks (CPU only): the library's 3-D geometry, post-process and
merge steps, and the oracle's restatement of them, against the goldens produced by RUNNING the
reference's own DddDetector / ExdetDetector methods and utils/ddd_utils.py
(tests/golden/gen_golden_tasks.py).  Bit-exact: it is float32 / float64 host arithmetic."""
import importlib.util
import os
import types

import numpy as np
import pytest
import torch

from centernet_amd import ddd_utils as U
from centernet_amd.detectors.ddd import DddDetector
from centernet_amd.detectors.exdet import ExdetDetector
from oracle import post_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("gen_golden_tasks", os.path.join(HERE, "golden", "gen_golden_tasks.py"))
GEN = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(GEN)
GOLD = np.load(os.path.join(HERE, "golden", "tasks_golden.npz"))
DEFAULT_CALIB = np.array([[707.0493, 0, 604.0814, 45.75831], [0, 707.0493, 180.5066, -0.3454157],
                          [0, 0, 1., 0.004981016]], dtype=np.float32)


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True)


def _ddd_meta(name):
    d, meta = GEN.ddd_inputs(name)
    if meta["calib"] is None:
        meta["calib"] = DEFAULT_CALIB
    return d, meta


@pytest.mark.parametrize("name", sorted(GEN.DDD_CASES))
def test_ddd_post_process_and_merge_equal_the_reference(name):
    d, meta = _ddd_meta(name)
    me = types.SimpleNamespace(opt=GEN.ddd_opt(), num_classes=3)
    per_class = DddDetector.post_process(me, torch.from_numpy(d.copy()), meta)
    assert me.this_calib is meta["calib"]
    for j in (1, 2, 3):
        assert _same(per_class[j], GOLD["ddd/%s/post/%d" % (name, j)]), (name, j)
    merged = DddDetector.merge_outputs(me, [per_class])
    for j in (1, 2, 3):
        assert _same(merged[j], GOLD["ddd/%s/merged/%d" % (name, j)]), (name, j)
    # the oracle's restatement: the same bits
    ref = post_oracle.ddd_results(d, meta, 3, GEN.DDD_OUT[1], GEN.DDD_OUT[0])
    for j in (1, 2, 3):
        assert _same(ref[j], GOLD["ddd/%s/post/%d" % (name, j)]), (name, j)
    ref = post_oracle.ddd_merge_outputs([ref], 3, 0.2)
    for j in (1, 2, 3):
        assert _same(ref[j], GOLD["ddd/%s/merged/%d" % (name, j)]), (name, j)


def test_ddd_rows_are_alpha_box_dims_location_yaw_score():
    d, meta = _ddd_meta("kitti_p2")
    me = types.SimpleNamespace(opt=GEN.ddd_opt(), num_classes=3)
    per_class = DddDetector.post_process(me, torch.from_numpy(d.copy()), meta)
    rows = per_class[1]
    assert rows.shape[1] == 13 and rows.dtype == np.float32 and per_class[2].shape == (0,)
    assert np.all(rows[:, 3] > rows[:, 1]) and np.all(rows[:, 4] > rows[:, 2])          # x2 > x1, y2 > y1
    assert np.all(np.abs(rows[:, 11]) <= np.float32(np.pi) * (1 + 1e-6))                # yaw wrapped
    # scores: the class's rows in decode order, i.e. descending
    assert np.all(np.diff(rows[:, 12]) <= 0)
    # the location projects back onto the predicted centre (minus the h / 2 shift to the bottom face)
    P = meta["calib"]
    loc = rows[:, 8:11].copy()
    loc[:, 1] -= rows[:, 5] / 2
    back = U.project_to_image(loc, P)
    sel = d[0, :, 17] == 0
    from centernet_amd.image import transform_preds
    centre = transform_preds(d[0, sel, 0:2], meta["c"], meta["s"], (GEN.DDD_OUT[1], GEN.DDD_OUT[0]))
    assert np.abs(back - centre).max() < 2e-2


def test_ddd_geometry_functions_equal_the_reference():
    P = np.array(GEN.KITTI_CALIB, dtype=np.float32)
    for i, (dim, loc, ry, px, depth, alpha) in enumerate(GEN.geometry_inputs()):
        assert _same(U.compute_box_3d(dim, loc, ry), GOLD["geo/%d/box3d" % i])
        assert _same(U.project_3d_bbox(loc, dim, ry, P), GOLD["geo/%d/box2d" % i])
        assert _same(U.compute_orientation_3d(dim, loc, ry), GOLD["geo/%d/orient" % i])
        assert _same(U.unproject_2d_to_3d(px, depth, P), GOLD["geo/%d/unproject" % i])
        assert _same(np.asarray(U.alpha2rot_y(alpha, px[0], P[0, 2], P[0, 0])), GOLD["geo/%d/rot_y" % i])
        assert _same(np.asarray(U.rot_y2alpha(ry, px[0], P[0, 2], P[0, 0])), GOLD["geo/%d/alpha" % i])
        locs, rot = U.ddd2locrot(px, alpha, dim, depth, P)
        assert _same(np.concatenate([locs, [rot]]).astype(np.float64), GOLD["geo/%d/locrot" % i])
        # properties: the two angle maps invert each other; the box is a rigid copy of its dimensions
        back = U.rot_y2alpha(U.alpha2rot_y(alpha, px[0], P[0, 2], P[0, 0]), px[0], P[0, 2], P[0, 0])
        assert abs(float(back) - float(alpha)) < 1e-5 or abs(abs(float(back) - float(alpha)) - 2 * np.pi) < 1e-5
        box = U.compute_box_3d(dim, loc, ry)
        assert abs(np.linalg.norm(box[0] - box[1]) - dim[1]) < 1e-4      # w
        assert abs(np.linalg.norm(box[0] - box[3]) - dim[2]) < 1e-4      # l
        assert abs(np.linalg.norm(box[0] - box[4]) - dim[0]) < 1e-4      # h, along y
        assert np.allclose(box[:4].mean(axis=0), loc, atol=1e-4)         # location = centre of the bottom face
    # the known answer printed by the reference file's own __main__ (ddd_utils.py:122-130)
    tl, br = np.array([712.40, 143.00], dtype=np.float32), np.array([810.73, 307.92], dtype=np.float32)
    ct = (tl + br) / 2
    got = U.alpha2rot_y(-0.20, ct[0], DEFAULT_CALIB[0, 2], DEFAULT_CALIB[0, 0])
    assert _same(np.asarray(got), GOLD["geo/main/rot_y"]) and abs(float(got) - 0.01) < 0.02


def _ddd_host_detector(extra=()):
    """A DddDetector without its network (the constructor needs the device): host methods only."""
    from centernet_amd.opts import opts
    opt = opts().init(["ddd"] + list(extra))
    det = DddDetector.__new__(DddDetector)
    det.opt, det.num_classes = opt, opt.num_classes
    det.mean = np.asarray(opt.mean, np.float32).reshape(1, 1, 3)
    det.std = np.asarray(opt.std, np.float32).reshape(1, 1, 3)
    det.calib = DEFAULT_CALIB
    return det, opt


@pytest.mark.parametrize("shape,extra", [((375, 1242), ()), ((370, 1224), ()), ((188, 621), ("--keep_res",)),
                                         ((64, 96), ("--input_h", "64", "--input_w", "128"))])
def test_ddd_pre_process_equals_the_oracle(shape, extra):
    """detectors/ddd.py:30-54 against oracle/pre_oracle.py (OpenCV's fixed-point warp restated, float32
    normalisation): identical bits; meta as the reference builds it (int32 extent, calib default)."""
    from oracle import pre_oracle
    det, opt = _ddd_host_detector(extra)
    image = np.random.RandomState(sum(shape)).randint(0, 256, shape + (3,)).astype(np.uint8)
    images, meta = det.pre_process(image, 1.0)
    ref, rmeta = pre_oracle.ddd_pre_process(image, opt.mean, opt.std, opt.input_h, opt.input_w, keep_res=opt.keep_res)
    assert images.dtype == torch.float32 and tuple(images.shape) == (1, 3, opt.input_h, opt.input_w)
    assert np.array_equal(images.numpy(), ref)
    assert meta["s"].dtype == np.int32 and np.array_equal(meta["s"], rmeta["s"]) and np.array_equal(meta["c"], rmeta["c"])
    assert (meta["out_height"], meta["out_width"]) == (opt.input_h // 4, opt.input_w // 4)
    assert meta["calib"] is DEFAULT_CALIB
    custom = det.pre_process(image, 1.0, GEN.KITTI_CALIB)[1]["calib"]
    assert custom.dtype == np.float32 and np.array_equal(custom, np.array(GEN.KITTI_CALIB, np.float32))
    # the float32 chain is NOT the other tasks' float64 one: some levels differ in the last bit
    from centernet_amd.image import normalize_chw_numpy
    lv = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)
    f32 = ((lv.astype(np.float32) / 255. - det.mean) / det.std).transpose(2, 0, 1)
    assert not np.array_equal(f32, normalize_chw_numpy(lv, opt.mean, opt.std))


def test_ddd_and_exdet_have_no_frame_pipeline():
    det, _ = _ddd_host_detector()
    with pytest.raises(NotImplementedError):
        det.run_frames([np.zeros((8, 8, 3), np.uint8)])


def _exdet_self():
    return types.SimpleNamespace(opt=types.SimpleNamespace(), num_classes=80, max_per_image=100)


@pytest.mark.parametrize("name", sorted(GEN.EXDET_CASES))
def test_exdet_post_process_and_merge_equal_the_reference(name):
    d, meta, scale = GEN.exdet_inputs(name)
    rows = ExdetDetector.post_process(_exdet_self(), torch.from_numpy(d.copy()), meta, scale)
    assert _same(rows, GOLD["exdet/%s/post" % name])
    assert _same(post_oracle.exdet_post_process(d, meta, scale), GOLD["exdet/%s/post" % name])
    for merge in (lambda r: ExdetDetector.merge_outputs(_exdet_self(), [r]),
                  lambda r: post_oracle.exdet_merge_outputs([r], 80)):
        merged = merge(rows.copy())
        assert sorted(merged) == list(range(1, 81))
        for j in range(1, 81):
            key = "exdet/%s/merged/%d" % (name, j)
            if key in GOLD.files:
                assert _same(merged[j], GOLD[key]), (name, j)
            else:
                assert merged[j].shape == (0, 5) and merged[j].dtype == np.float32
    if name == "crowded":   # the threshold cut was exercised
        n = sum(len(GOLD[k]) for k in GOLD.files if k.startswith("exdet/crowded/merged/"))
        assert 100 <= n < 2 * 400


def test_exdet_two_scales_merge_equals_the_reference():
    per_scale = []
    for name in ("flip_512", "flip_half"):
        d, meta, scale = GEN.exdet_inputs(name)
        per_scale.append(ExdetDetector.post_process(_exdet_self(), torch.from_numpy(d.copy()), meta, scale))
    for merged in (ExdetDetector.merge_outputs(_exdet_self(), [r.copy() for r in per_scale]),
                   post_oracle.exdet_merge_outputs([r.copy() for r in per_scale], 80)):
        seen = 0
        for j in range(1, 81):
            key = "exdet/two_scales/merged/%d" % j
            if key in GOLD.files:
                assert _same(merged[j], GOLD[key]), j
                seen += len(merged[j])
            else:
                assert len(merged[j]) == 0
        assert seen >= 100


def test_exdet_second_half_is_unmirrored_and_only_the_box_is_moved():
    d, meta, scale = GEN.exdet_inputs("flip_512")
    rows = ExdetDetector.post_process(_exdet_self(), torch.from_numpy(d.copy()), meta, scale)
    n = d.shape[1]
    assert rows.shape == (2 * n, 14)
    # 512 x 512 frame on a 128 x 128 grid: the inverse map is x 4
    assert np.allclose(rows[:n, 0:4], d[0, :, 0:4] * 4, atol=1e-3)
    assert np.allclose(rows[n:, 0], (128 - d[1, :, 2]) * 4, atol=1e-3) and np.allclose(rows[n:, 2], (128 - d[1, :, 0]) * 4, atol=1e-3)
    assert np.array_equal(rows[:, 4:], d.reshape(-1, 14)[:, 4:])        # score, extreme points, class: untouched


def test_exdet_guard_rails():
    """--K above what the K^4 grouping kernel takes is refused before any device work."""
    from centernet_amd.opts import opts
    # a command line that does not pass --K runs exdet at ExtremeNet's own K = 40 (the reference default of 100
    # means 10^8 groupings per image) -- an explicit --K is taken as given, and refused above 64
    assert opts().init(["exdet", "--arch", "hourglass"]).K == 40
    assert opts().init(["exdet", "--K", "100"]).K == 100 and opts().init(["ctdet"]).K == 100
    with pytest.raises(ValueError):
        ExdetDetector(opts().init(["exdet", "--arch", "hourglass", "--K", "100"]))
    with pytest.raises(ValueError):
        ExdetDetector(opts().init(["exdet", "--agnostic_ex", "--K", "65"]))


def test_kitti_result_files_equal_the_reference(tmp_path):
    """datasets/dataset/kitti.py:68-82 (KITTI.save_results run on the ddd goldens): same files, same bytes."""
    import json
    from centernet_amd import results as R
    golden = json.load(open(os.path.join(HERE, "golden", "tasks_kitti_golden.json")))
    inp = GEN.kitti_results_inputs(GOLD)
    R.save_results_kitti(inp, str(tmp_path))
    got = {n: open(tmp_path / "results" / n).read() for n in sorted(os.listdir(tmp_path / "results"))}
    assert got == golden and sorted(got) == ["000007.txt", "000123.txt"]
    first = got["000007.txt"].splitlines()[0].split()
    assert first[0] in R.KITTI_CLASS_NAMES[1:] and first[1:3] == ["0.0", "0"] and len(first) == 3 + 13
    assert all(len(v.split(".")[1]) == 2 for v in first[3:])
