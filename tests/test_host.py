"""Host-side mirror of the reference interface: opts, geometry, post-process (CPU only)."""
import numpy as np
import pytest

from centernet_amd import image as img
from centernet_amd.opts import opts
from centernet_amd.post_process import ctdet_post_process, multi_pose_post_process
from oracle import post_oracle


def test_opts_init_ctdet_defaults():
    o = opts().init(["ctdet", "--arch", "resdcn_18"])
    assert o.heads == {"hm": 80, "wh": 2, "reg": 2} and o.head_conv == 64
    assert (o.input_h, o.input_w, o.output_h, o.output_w) == (512, 512, 128, 128)
    assert o.K == 100 and o.gpus == [0] and o.test_scales == [1.0] and o.fix_res
    assert o.mean == [0.408, 0.447, 0.470] and o.num_classes == 80 and o.down_ratio == 4
    o = opts().init(["ctdet", "--arch", "dla_34", "--cat_spec_wh", "--not_reg_offset", "--gpus", "-1"])
    assert o.heads == {"hm": 80, "wh": 160} and o.head_conv == 256 and o.gpus == [-1]
    o = opts().init(["multi_pose", "--arch", "hourglass", "--keep_res", "--test_scales", "0.5,1,2"])
    assert o.heads == {"hm": 1, "wh": 2, "hps": 34, "reg": 2, "hm_hp": 17, "hp_offset": 2}
    assert o.pad == 127 and o.num_stacks == 2 and not o.fix_res and o.test_scales == [0.5, 1.0, 2.0]
    # the ddd / exdet tasks (opts.py:299-312,341-353): kitti geometry, the 3-D heads, the five extreme-point maps
    o = opts().init(["ddd"])
    assert o.heads == {"hm": 3, "dep": 1, "rot": 8, "dim": 3, "wh": 2, "reg": 2} and o.arch == "dla_34"
    assert (o.input_h, o.input_w, o.output_h, o.output_w) == (384, 1280, 96, 320) and o.dataset == "kitti"
    assert o.mean == [0.485, 0.456, 0.406] and o.std == [0.229, 0.224, 0.225] and o.peak_thresh == 0.2
    assert opts().init(["ddd", "--not_reg_bbox", "--not_reg_offset"]).heads == {"hm": 3, "dep": 1, "rot": 8, "dim": 3}
    o = opts().init(["exdet", "--arch", "hourglass"])
    assert o.heads == {"hm_t": 80, "hm_l": 80, "hm_b": 80, "hm_r": 80, "hm_c": 80,
                       "reg_t": 2, "reg_l": 2, "reg_b": 2, "reg_r": 2}
    assert (o.scores_thresh, o.center_thresh, o.aggr_weight) == (0.1, 0.1, 0.0)
    o = opts().init(["exdet", "--agnostic_ex", "--not_reg_offset"])
    assert o.heads == {"hm_t": 1, "hm_l": 1, "hm_b": 1, "hm_r": 1, "hm_c": 80}
    with pytest.raises(NotImplementedError):
        opts().init(["segmentation"])


def test_affine_identity_for_benchmark_configuration():
    """512x512 input with fix_res: c=(256,256), s=512 -> input warp is the identity and the
    output-grid -> image map is x4 (SURVEY.md section 7 'No cv2')."""
    c = np.array([256., 256.], np.float32)
    t = img.get_affine_transform(c, 512.0, 0, [512, 512])
    assert np.allclose(t, [[1, 0, 0], [0, 1, 0]], atol=1e-9)
    ti = img.get_affine_transform(c, 512.0, 0, [128, 128], inv=1)
    assert np.allclose(ti, [[4, 0, 0], [0, 4, 0]], atol=1e-9)


@pytest.mark.parametrize("case", [((640., 480.), 640.0, (128, 128)), ((333., 250.), 500.0, (128, 96)),
                                  ((320., 240.), np.array([672., 512.], np.float32), (168, 128))])
def test_transform_preds_matches_oracle(case):
    (w, h), s, out = case
    c = np.array([w / 2, h / 2], np.float32)
    pts = np.random.RandomState(0).uniform(0, 120, (50, 2)).astype(np.float32)
    got = img.transform_preds(pts, c, s, out)
    ref = post_oracle.transform_preds(pts, c, s, out)
    assert np.allclose(got, ref, rtol=0, atol=1e-4)


@pytest.mark.parametrize("rot", [0, 30, -75.5, 180])
@pytest.mark.parametrize("inv", [0, 1])
def test_affine_similarity_matches_three_point_solve(rot, inv):
    """The closed-form similarity == the reference's three-point construction solved numerically
    (oracle), for rotations and shifts too (training-time augmentation values)."""
    c = np.array([301.5, 222.25], np.float32)
    for s in (417.0, np.array([640., 480.], np.float32)):
        for shift in ((0.0, 0.0), (0.1, -0.05)):
            got = img.get_affine_transform(c, s, rot, [128, 96], shift=np.array(shift, np.float32), inv=inv)
            ref = post_oracle.get_affine_transform(c, s, rot, [128, 96], shift=np.array(shift, np.float32), inv=inv)
            assert got.shape == (2, 3) and got.dtype == np.float64
            assert np.allclose(got, ref, rtol=1e-6, atol=2e-4), (rot, inv)


def test_multi_pose_post_process_matches_oracle():
    rs = np.random.RandomState(4)
    dets = rs.uniform(0, 128, (2, 30, 40)).astype(np.float32)
    c = [np.array([250., 187.5], np.float32), np.array([320., 240.], np.float32)]
    s = [500.0, np.array([672., 512.], np.float32)]
    got = multi_pose_post_process(dets.copy(), c, s, 128, 128)
    ref = post_oracle.multi_pose_post_process(dets.copy(), c, s, 128, 128)
    for g, r in zip(got, ref):
        assert list(g.keys()) == [1] and list(r.keys()) == [1]
        a, b = np.array(g[1], np.float32), np.array(r[1], np.float32)
        assert a.shape == b.shape == (30, 39)
        assert np.array_equal(a[:, 4], b[:, 4])
        assert (np.abs(a - b) <= np.spacing(np.abs(b))).all()


def test_post_process_does_not_modify_its_input_and_keeps_row_order():
    rs = np.random.RandomState(6)
    dets = np.concatenate([rs.uniform(0, 128, (1, 40, 4)), rs.uniform(0, 1, (1, 40, 1)),
                           rs.randint(0, 3, (1, 40, 1))], axis=2).astype(np.float32)
    keep = dets.copy()
    out = ctdet_post_process(dets, [np.array([256., 256.], np.float32)], [512.0], 128, 128, 3)[0]
    assert np.array_equal(dets, keep)
    for j in range(3):
        want = keep[0, keep[0, :, 5] == j, 4]
        assert np.array_equal(np.array(out[j + 1], np.float32).reshape(-1, 5)[:, 4], want)


def test_ctdet_post_process_matches_oracle():
    rs = np.random.RandomState(1)
    dets = np.concatenate([rs.uniform(0, 128, (2, 100, 4)), rs.uniform(0, 1, (2, 100, 1)),
                           rs.randint(0, 80, (2, 100, 1))], axis=2).astype(np.float32)
    c = [np.array([250., 187.5], np.float32)] * 2
    s = [500.0] * 2
    got = ctdet_post_process(dets.copy(), c, s, 128, 128, 80)
    ref = post_oracle.ctdet_post_process(dets.copy(), c, s, 128, 128, 80)
    for g, r in zip(got, ref):
        assert set(g) == set(range(1, 81))
        for j in range(1, 81):
            assert np.allclose(np.array(g[j]).reshape(-1, 5), np.array(r[j]).reshape(-1, 5), atol=1e-4)


def test_multi_pose_post_process_shapes():
    rs = np.random.RandomState(2)
    dets = rs.uniform(0, 128, (1, 100, 40)).astype(np.float32)
    out = multi_pose_post_process(dets.copy(), [np.array([256., 256.], np.float32)], [512.0], 128, 128)
    assert list(out[0].keys()) == [1] and np.array(out[0][1]).shape == (100, 39)
    assert np.allclose(np.array(out[0][1])[:, :4], dets[0, :, :4] * 4, atol=1e-3)


def test_warp_affine_identity_and_shift():
    rs = np.random.RandomState(3)
    im = rs.randint(0, 255, (40, 60, 3)).astype(np.uint8)
    assert np.array_equal(img.warp_affine(im, np.array([[1, 0, 0], [0, 1, 0.]]), (60, 40)), im)
    sh = img.warp_affine(im, np.array([[1, 0, 2.], [0, 1, 3.]]), (60, 40))
    assert np.array_equal(sh[3:, 2:], im[:-3, :-2]) and sh[:3].max() == 0
    up = img.resize_bilinear(im, (120, 80))
    assert up.shape == (80, 120, 3)


def test_detector_rejects_cpu_mode():
    from centernet_amd.detectors import detector_factory
    from centernet_amd.native import NativeError
    o = opts().init(["ctdet", "--arch", "res_18", "--gpus", "-1"])
    with pytest.raises(NativeError):
        detector_factory[o.task](o)


@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("ncol", [5, 39])
def test_soft_nms_matches_restatement(method, ncol):
    from centernet_amd.soft_nms import soft_nms, soft_nms_39
    rs = np.random.RandomState(method * 7 + ncol)
    n = 60
    xy = rs.uniform(0, 200, (n, 2))
    wh = rs.uniform(10, 80, (n, 2))
    boxes = np.zeros((n, ncol), np.float32)
    boxes[:, 0:2] = xy
    boxes[:, 2:4] = xy + wh
    boxes[:, 4] = rs.uniform(0.001, 1, n)
    if ncol == 39:
        boxes[:, 5:] = rs.uniform(0, 200, (n, 34))
    a, b = boxes.copy(), boxes.copy()
    keep_a = (soft_nms if ncol == 5 else soft_nms_39)(a, Nt=0.5, method=method)
    keep_b = post_oracle.soft_nms(b, Nt=0.5, method=method)
    assert keep_a == keep_b
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_soft_nms_known_answer():
    """Two identical boxes, hard NMS: the second is discarded; gaussian: decayed by e^-2."""
    from centernet_amd.soft_nms import soft_nms
    b = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 9, 0.8], [100, 100, 109, 109, 0.7]], np.float32)
    a = b.copy()
    assert soft_nms(a, Nt=0.5, method=0) == [0, 1] and a[0, 4] == np.float32(0.9) and a[1, 4] == np.float32(0.7)
    g = b.copy()
    assert soft_nms(g, Nt=0.5, method=2) == [0, 1, 2]
    assert abs(g[np.argmin(np.abs(g[:, 4] - 0.8 * np.exp(-2.0))), 4] - 0.8 * np.exp(-2.0)) < 1e-6


def test_ctdet_results_batch_equals_per_image_loop():
    """Vectorised batch tail == post_process + merge_outputs per image (bit-identical rows)."""
    from centernet_amd.post_process import ctdet_results_batch
    from oracle import post_oracle
    rng = np.random.RandomState(3)
    for K, max_per in ((100, 100), (120, 100)):
        B = 3
        dets = np.zeros((B, K, 6), np.float32)
        dets[:, :, :4] = rng.uniform(0, 128, (B, K, 4))
        dets[:, :, 4] = np.sort(rng.uniform(0, 1, (B, K)), axis=1)[:, ::-1]
        dets[:, :, 5] = rng.randint(0, 80, (B, K))
        metas = [{'c': np.array([250., 187.5], np.float32), 's': 500.0, 'out_height': 128,
                  'out_width': 128} for _ in range(B)]
        got = ctdet_results_batch(dets.copy(), metas, 80, scale=1, max_per_image=max_per)
        for i in range(B):
            ref = post_oracle.ctdet_results(dets[i:i + 1].copy(), metas[i], 80, scale=1,
                                            max_per_image=max_per)
            for j in range(1, 81):
                assert got[i][j].dtype == np.float32 and got[i][j].shape == ref[j].shape
                assert np.array_equal(got[i][j][:, 4], ref[j][:, 4])
                assert (np.abs(got[i][j] - ref[j]) <= np.spacing(np.abs(ref[j]))).all()


def test_host_warp_routine_speed_and_oracle():
    """cn_warp_affine_u8_host (C, in the library) == the OpenCV restatement of oracle/pre_oracle.py
    on a full 640 x 480 -> 512 x 512 frame, and it is the fast form of it."""
    import time
    from centernet_amd.image import get_affine_transform, warp_affine
    from oracle import pre_oracle as P
    rng = np.random.RandomState(2)
    big = rng.randint(0, 256, (480, 640, 3)).astype(np.uint8)
    M = get_affine_transform(np.array([320., 240.], np.float32), 640.0, 0, (512, 512))
    t0 = time.perf_counter(); a = warp_affine(big, M, (512, 512)); t1 = time.perf_counter()
    b = P.cv_warp_affine_u8(big, M, (512, 512)); t2 = time.perf_counter()
    assert np.array_equal(a, b)
    assert (t1 - t0) < (t2 - t1), "the C routine should be faster than the numpy statement"


def test_host_normalise_routine_equals_the_numpy_definition():
    """cn_normalize_u8_chw_f32_host == ((img / 255. - mean) / std).astype(float32) + CHW, bit for bit
    (every uint8 level occurs in the larger images)."""
    from centernet_amd.image import normalize_chw, normalize_chw_numpy
    rng = np.random.RandomState(4)
    mean = np.array([0.408, 0.447, 0.470], np.float32)
    std = np.array([0.289, 0.274, 0.278], np.float32)
    for shape in ((1, 1, 3), (37, 53, 3), (128, 96, 3)):
        img = rng.randint(0, 256, shape).astype(np.uint8)
        a, b = normalize_chw(img, mean.reshape(1, 1, 3), std.reshape(1, 1, 3)), normalize_chw_numpy(img, mean, std)
        assert a.dtype == np.float32 and a.shape == (3,) + shape[:2]
        assert np.array_equal(a, np.ascontiguousarray(b))


def test_ctdet_results_batch_with_mixed_frame_geometries():
    """Images of different geometry in one batch: every distinct (centre, extent, grid) gets its
    own inverse map, images sharing one go through it together."""
    from centernet_amd.post_process import ctdet_results_batch
    from oracle import post_oracle
    rng = np.random.RandomState(8)
    B, K = 5, 100
    dets = np.zeros((B, K, 6), np.float32)
    dets[:, :, :4] = rng.uniform(0, 128, (B, K, 4))
    dets[:, :, 4] = np.sort(rng.uniform(0, 1, (B, K)), axis=1)[:, ::-1]
    dets[:, :, 5] = rng.randint(0, 80, (B, K))
    a = {'c': np.array([250., 187.5], np.float32), 's': 500.0, 'out_height': 128, 'out_width': 128}
    b = {'c': np.array([320., 240.], np.float32), 's': np.array([640., 480.], np.float32),
         'out_height': 128, 'out_width': 128}
    metas = [a, b, dict(a), b, a]
    got = ctdet_results_batch(dets.copy(), metas, 80, scale=2.0, max_per_image=100)
    for i in range(B):
        ref = post_oracle.ctdet_results(dets[i:i + 1].copy(), metas[i], 80, scale=2.0, max_per_image=100)
        for j in range(1, 81):
            assert got[i][j].shape == ref[j].shape
            assert (np.abs(got[i][j] - ref[j]) <= np.spacing(np.abs(ref[j]))).all()


def test_detector_post_process_and_merge_match_the_oracle_tail():
    """CtdetDetector.post_process + merge_outputs (the single-image API, ctdet.py:47-73) against
    the oracle's restatement: one scale, two scales with soft-NMS, and more detections than the
    cap.  The detector object is built without a model: only the host tail is exercised."""
    import types
    import torch
    from centernet_amd.detectors.ctdet import CtdetDetector
    from oracle import post_oracle
    rng = np.random.RandomState(11)
    meta = {'c': np.array([250., 187.5], np.float32), 's': 500.0, 'out_height': 128, 'out_width': 128}

    def dets_for(K):
        d = np.zeros((1, K, 6), np.float32)
        d[0, :, :4] = rng.uniform(0, 128, (K, 4))
        d[0, :, 4] = np.sort(rng.uniform(0, 1, K))[::-1]
        d[0, :, 5] = rng.randint(0, 80, K)
        return d
    for K, scales, nms in ((100, [1.0], False), (120, [1.0], False), (100, [1.0, 2.0], False), (100, [1.0], True)):
        det = object.__new__(CtdetDetector)
        det.num_classes, det.max_per_image, det.scales = 80, 100, scales
        det.opt = types.SimpleNamespace(num_classes=80, nms=nms)
        per_scale, ref_scale = [], []
        for sc in scales:
            d = dets_for(K)
            per_scale.append(det.post_process(torch.from_numpy(d.copy()), meta, sc))
            ref_scale.append(post_oracle.ctdet_post_process_scale(d.copy(), meta, 80, sc))
        got = det.merge_outputs(per_scale)
        ref = post_oracle.ctdet_merge_outputs(ref_scale, 80, len(scales), nms=nms, max_per_image=100)
        for j in range(1, 81):
            assert got[j].dtype == np.float32 and got[j].shape == ref[j].shape, (K, scales, nms, j)
            assert (np.abs(got[j] - ref[j]) <= 2 * np.spacing(np.abs(ref[j]))).all()


def test_load_model_tolerant_like_the_reference(tmp_path, capsys):
    """model.py:31-67 behaviour: DataParallel prefixes stripped, shape mismatches skipped with a
    message, unknown keys dropped, missing keys keep the model's value; save_model round trip."""
    import torch
    from centernet_amd import synth
    from centernet_amd.model import create_model, load_model, save_model
    heads = {'hm': 80, 'wh': 2, 'reg': 2}
    src = create_model('res_18', heads, 64)
    synth.fill_state_dict_(src, 5)
    sd = {('module.' + k): v.clone() for k, v in src.state_dict().items()}
    sd['module.hm.2.weight'] = torch.zeros(3, 64, 1, 1)          # other --num_classes: skipped
    sd['module.not_in_model'] = torch.zeros(1)                    # dropped
    missing = 'module.wh.2.bias'
    del sd[missing]                                                # kept from the target model
    path = str(tmp_path / 'ckpt.pth')
    torch.save({'epoch': 7, 'state_dict': sd}, path)
    dst = create_model('res_18', heads, 64)
    synth.fill_state_dict_(dst, 9)
    keep_hm = dst.state_dict()['hm.2.weight'].clone()
    keep_wh = dst.state_dict()['wh.2.bias'].clone()
    load_model(dst, path)
    out = capsys.readouterr().out
    assert 'epoch 7' in out and 'Skip loading parameter hm.2.weight' in out
    assert 'Drop parameter not_in_model' in out and 'No param wh.2.bias' in out
    got = dst.state_dict()
    assert torch.equal(got['hm.2.weight'], keep_hm) and torch.equal(got['wh.2.bias'], keep_wh)
    for k, v in src.state_dict().items():
        if k not in ('hm.2.weight', 'wh.2.bias'):
            assert torch.equal(got[k], v), k
    save_model(str(tmp_path / 'again.pth'), 3, dst)
    again = torch.load(str(tmp_path / 'again.pth'), weights_only=False)
    assert again['epoch'] == 3 and set(again['state_dict']) == set(got)
    with pytest.raises(KeyError):
        create_model('dlav0_34', heads, 64)


def test_load_model_resumes_optimizer_and_refuses_pickled_code(tmp_path, capsys):
    """model.py:69-84: with an optimizer the call returns (model, optimizer, start_epoch) and a
    resume restores the state and the step-decayed rate; checkpoints that would execute pickled
    code are refused unless CENTERNET_UNSAFE_LOAD=1."""
    import torch
    from centernet_amd.model import create_model, load_model, save_model
    heads = {'hm': 80, 'wh': 2, 'reg': 2}
    m = create_model('res_18', heads, 64)
    opt = torch.optim.Adam(m.parameters(), lr=1.25e-4)
    path = str(tmp_path / 'with_opt.pth')
    save_model(path, 95, m, opt)
    m2 = create_model('res_18', heads, 64)
    opt2 = torch.optim.Adam(m2.parameters(), lr=1.0)
    got = load_model(m2, path, opt2, resume=True, lr=1.25e-4, lr_step=[90, 120])
    assert got[0] is m2 and got[1] is opt2 and got[2] == 95
    assert all(abs(g['lr'] - 1.25e-5) < 1e-12 for g in opt2.param_groups)     # one step behind
    assert 'Resumed optimizer with start lr' in capsys.readouterr().out
    _, _, e0 = load_model(m2, path, opt2, resume=False, lr=1.25e-4, lr_step=[90, 120])
    assert e0 == 0

    class Boom(object):
        def __reduce__(self):
            return (print, ("pickled code ran",))
    bad = str(tmp_path / 'bad.pth')
    torch.save({'epoch': 1, 'state_dict': m.state_dict(), 'extra': Boom()}, bad)
    with pytest.raises(RuntimeError, match="weights_only"):
        load_model(m2, bad)
    assert "pickled code ran" not in capsys.readouterr().out
