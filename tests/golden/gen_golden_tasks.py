#!/usr/bin/env python
"""Golden vectors for the host side of the ddd and exdet tasks, produced by RUNNING the reference:

* ``DddDetector.post_process`` / ``merge_outputs`` (src/lib/detectors/ddd.py:75-88) -- i.e. the
  reference's ``ddd_post_process`` (utils/post_process.py:10-86) and ``utils/ddd_utils.py``;
* the numeric functions of ``utils/ddd_utils.py`` on their own, and the file's own ``__main__``
  known-answer (ddd_utils.py:122-130);
* ``ExdetDetector.post_process`` / ``merge_outputs`` (src/lib/detectors/exdet.py:86-123), with the
  ``soft_nms`` the class forgets to import bound to the reference's own cython build
  (oracle/_ref, external/nms.pyx).

The two detector modules are imported from where they lie with their missing third-party imports
stubbed: ``cv2`` (only ``getAffineTransform`` is reached: the restatement of oracle/pre_oracle.py,
pinned in tests/test_oracle_pre.py), ``progress``, ``utils.debugger`` and the ``BaseDetector`` base
class (torchvision / DCNv2 imports); the methods are then called as plain functions on a stand-in
``self``.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_tasks.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

KITTI_CALIB = [[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791],
               [0.0, 0.0, 1.0, 0.002745884]]          # a training-set P2, the values of a KITTI calib file

DDD_CASES = {
    # name: (K, frame (h, w), classes present, calib or None = the detector's default, with wh)
    "kitti_default_calib": (40, (375, 1242), (0, 1, 2), None),
    "kitti_p2": (40, (370, 1224), (0, 2), KITTI_CALIB),            # class 2 (index 1) never fires: the (0,) case
    "few": (3, (375, 1242), (1,), KITTI_CALIB),
}
DDD_OUT = (96, 320)      # output grid of the 384 x 1280 input


def ddd_inputs(name):
    """(1, K, 18) rows shaped like ``ddd_decode``'s, the meta of DddDetector.pre_process."""
    K, (h, w), classes, calib = DDD_CASES[name]
    rs = np.random.RandomState(4000 + sum(map(ord, name)))
    d = np.zeros((1, K, 18), np.float32)
    d[0, :, 0] = rs.uniform(0, DDD_OUT[1], K)
    d[0, :, 1] = rs.uniform(0, DDD_OUT[0], K)
    d[0, :, 2] = np.sort(rs.uniform(0.02, 0.95, K))[::-1]
    d[0, :, 3:11] = rs.normal(0, 1, (K, 8))
    d[0, :, 11] = rs.uniform(1, 60, K)
    d[0, :, 12:15] = rs.uniform(0.5, 4, (K, 3))
    d[0, :, 15:17] = rs.uniform(2, 60, (K, 2))
    d[0, :, 17] = rs.choice(classes, K)
    meta = {"c": np.array([w / 2, h / 2], dtype=np.float32), "s": np.array([w, h], dtype=np.int32),
            "out_height": DDD_OUT[0], "out_width": DDD_OUT[1],
            "calib": None if calib is None else np.array(calib, dtype=np.float32)}
    return d, meta


def ddd_opt():
    return types.SimpleNamespace(output_w=DDD_OUT[1], output_h=DDD_OUT[0], num_classes=3, peak_thresh=0.2)


def geometry_inputs():
    """(dim (h, w, l), location, rotation_y, pixel, depth, alpha) tuples for the ddd_utils functions."""
    rs = np.random.RandomState(77)
    out = []
    for _ in range(6):
        out.append((rs.uniform(0.5, 4, 3).astype(np.float32), rs.uniform(-20, 40, 3).astype(np.float32),
                    np.float32(rs.uniform(-3.5, 3.5)), rs.uniform(0, 1242, 2).astype(np.float32),
                    np.float32(rs.uniform(2, 70)), np.float32(rs.uniform(-3.2, 3.2))))
    return out


EXDET_CASES = {
    # name: (rows per image of the decode, frame (h, w), test scale, classes)
    "flip_512": (60, (512, 512), 1.0, (0, 3, 17, 79)),
    "flip_half": (60, (480, 640), 0.5, (0, 3, 17, 79)),
    "crowded": (400, (427, 640), 1.0, (5, 6)),           # more than max_per_image rows survive: the threshold cut
}


def exdet_inputs(name):
    """(2, n, 14) rows shaped like ``exct_decode``'s for [frame, mirrored frame], the meta of
    BaseDetector.pre_process with fix_res (out 128 x 128), the test scale."""
    n, (h, w), scale, classes = EXDET_CASES[name]
    rs = np.random.RandomState(5000 + sum(map(ord, name)))
    d = np.zeros((2, n, 14), np.float32)
    x1 = rs.uniform(0, 100, (2, n))
    y1 = rs.uniform(0, 100, (2, n))
    d[:, :, 0], d[:, :, 1] = x1, y1
    d[:, :, 2] = x1 + rs.uniform(2, 28, (2, n))
    d[:, :, 3] = y1 + rs.uniform(2, 28, (2, n))
    sc = np.sort(rs.uniform(-0.6, 0.9, (2, n)), axis=1)[:, ::-1]     # rejected groupings carry scores <= 0
    d[:, :, 4] = sc
    d[:, :, 5:13] = rs.uniform(0, 128, (2, n, 8))
    d[:, :, 13] = rs.choice(classes, (2, n))
    sh, sw = int(h * scale), int(w * scale)
    meta = {"c": np.array([sw / 2., sh / 2.], dtype=np.float32), "s": max(h, w) * 1.0,
            "out_height": 128, "out_width": 128}
    return d, meta, scale


def kitti_results_inputs(gold):
    """{image id: run()['results']} for the KITTI writer, from the ddd goldens (an image with an empty class)"""
    return {7: {j: np.asarray(gold["ddd/kitti_default_calib/merged/%d" % j]) for j in (1, 2, 3)},
            123: {j: np.asarray(gold["ddd/kitti_p2/merged/%d" % j]) for j in (1, 2, 3)}}


def _import_reference_detectors():
    import importlib
    from oracle import pre_oracle, ref
    lib = "/root/reference/src/lib"
    sys.path.insert(0, lib)
    cv2 = types.ModuleType("cv2")
    cv2.getAffineTransform = lambda src, dst: pre_oracle.cv_get_affine_transform(np.float32(src), np.float32(dst))
    progress, bar = types.ModuleType("progress"), types.ModuleType("progress.bar")
    bar.Bar = object
    progress.bar = bar
    debugger = types.ModuleType("utils.debugger")
    debugger.Debugger = object
    pkg = types.ModuleType("detectors")
    pkg.__path__ = [os.path.join(lib, "detectors")]
    base = types.ModuleType("detectors.base_detector")
    base.BaseDetector = object
    sys.modules.update({"cv2": cv2, "progress": progress, "progress.bar": bar, "utils.debugger": debugger,
                        "detectors": pkg, "detectors.base_detector": base,
                        "_init_paths": types.ModuleType("_init_paths")})
    ddd = importlib.import_module("detectors.ddd")
    exdet = importlib.import_module("detectors.exdet")
    exdet.soft_nms = ref.soft_nms          # the name exdet.py:110 uses without importing it
    ddd_utils = importlib.import_module("utils.ddd_utils")
    return ddd, exdet, ddd_utils


def main():
    import torch
    ddd, exdet, U = _import_reference_detectors()
    out = {}
    default_calib = np.array([[707.0493, 0, 604.0814, 45.75831], [0, 707.0493, 180.5066, -0.3454157],
                              [0, 0, 1., 0.004981016]], dtype=np.float32)          # ddd.py:25-27
    for name in DDD_CASES:
        d, meta = ddd_inputs(name)
        if meta["calib"] is None:
            meta["calib"] = default_calib
        me = types.SimpleNamespace(opt=ddd_opt(), num_classes=3)
        per_class = ddd.DddDetector.post_process(me, torch.from_numpy(d.copy()), meta)
        for j in (1, 2, 3):
            out["ddd/%s/post/%d" % (name, j)] = np.asarray(per_class[j])
        merged = ddd.DddDetector.merge_outputs(me, [{j: np.asarray(per_class[j]).copy() for j in per_class}])
        for j in (1, 2, 3):
            out["ddd/%s/merged/%d" % (name, j)] = np.asarray(merged[j])
    P = np.array(KITTI_CALIB, dtype=np.float32)
    for i, (dim, loc, ry, px, depth, alpha) in enumerate(geometry_inputs()):
        out["geo/%d/box3d" % i] = U.compute_box_3d(dim, loc, ry)
        out["geo/%d/box2d" % i] = U.project_3d_bbox(loc, dim, ry, P)
        out["geo/%d/orient" % i] = U.compute_orientation_3d(dim, loc, ry)
        out["geo/%d/unproject" % i] = U.unproject_2d_to_3d(px, depth, P)
        out["geo/%d/rot_y" % i] = np.asarray(U.alpha2rot_y(alpha, px[0], P[0, 2], P[0, 0]))
        out["geo/%d/alpha" % i] = np.asarray(U.rot_y2alpha(ry, px[0], P[0, 2], P[0, 0]))
        locs, rot = U.ddd2locrot(px, alpha, dim, depth, P)
        out["geo/%d/locrot" % i] = np.concatenate([locs, [rot]]).astype(np.float64)
    # the file's own known answer (ddd_utils.py:122-130): prints alpha2rot_y of this box
    tl, br = np.array([712.40, 143.00], dtype=np.float32), np.array([810.73, 307.92], dtype=np.float32)
    ct = (tl + br) / 2
    out["geo/main/rot_y"] = np.asarray(U.alpha2rot_y(-0.20, ct[0], default_calib[0, 2], default_calib[0, 0]))

    per_scale = {}
    for name in EXDET_CASES:
        d, meta, scale = exdet_inputs(name)
        me = types.SimpleNamespace(opt=types.SimpleNamespace(), num_classes=80, max_per_image=100)
        rows = exdet.ExdetDetector.post_process(me, torch.from_numpy(d.copy()), meta, scale)
        out["exdet/%s/post" % name] = rows
        per_scale[name] = rows
        merged = exdet.ExdetDetector.merge_outputs(me, [rows.copy()])
        for j, v in merged.items():
            if len(v):
                out["exdet/%s/merged/%d" % (name, j)] = v
    me = types.SimpleNamespace(opt=types.SimpleNamespace(), num_classes=80, max_per_image=100)
    merged = exdet.ExdetDetector.merge_outputs(me, [per_scale["flip_512"].copy(), per_scale["flip_half"].copy()])
    for j, v in merged.items():
        if len(v):
            out["exdet/two_scales/merged/%d" % j] = v
    # the KITTI result files (datasets/dataset/kitti.py:68-82): one text file per image, written by the reference's
    # own KITTI.save_results from the ddd results above
    import importlib.util
    import json
    import tempfile
    stub = types.ModuleType("pycocotools")
    stub.coco = types.ModuleType("pycocotools.coco")
    sys.modules.update({"pycocotools": stub, "pycocotools.coco": stub.coco})
    spec = importlib.util.spec_from_file_location("ref_kitti", "/root/reference/src/lib/datasets/dataset/kitti.py")
    kitti = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kitti)
    results = kitti_results_inputs(out)
    me = types.SimpleNamespace(class_name=['__background__', 'Pedestrian', 'Car', 'Cyclist'])       # kitti.py:35-36
    files = {}
    with tempfile.TemporaryDirectory() as tmp:
        kitti.KITTI.save_results(me, results, tmp)
        for name in sorted(os.listdir(os.path.join(tmp, "results"))):
            files[name] = open(os.path.join(tmp, "results", name)).read()
    with open(os.path.join(HERE, "tasks_kitti_golden.json"), "w") as f:
        json.dump(files, f)
    print("kitti files:", {k: v.count("\n") for k, v in files.items()})
    np.savez_compressed(os.path.join(HERE, "tasks_golden.npz"), **out)
    print(len(out), "arrays;", {k: v.shape for k, v in list(out.items())[:8]})


if __name__ == "__main__":
    main()
