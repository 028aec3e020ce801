"""Dataset descriptors (centernet_amd/datasets.py, CPU only): what test.py reads off ``dataset_factory
[opt.dataset]`` -- against the attributes of the reference's own dataset classes (imported from where they
lie when /root/reference is present; the values are also spelled out here) and the result-writer goldens."""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import pytest

from centernet_amd import datasets as D
from centernet_amd.opts import opts

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/lib/datasets/dataset"


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden_results", os.path.join(HERE, "golden", "gen_golden_results.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_factory_and_the_values_the_detectors_use():
    assert sorted(D.dataset_factory) == ["coco", "coco_hp", "kitti", "pascal"]
    assert (D.COCO.num_classes, D.COCOHP.num_classes, D.PascalVOC.num_classes, D.KITTI.num_classes) == (80, 1, 20, 3)
    assert D.COCO.default_resolution == [512, 512] and D.PascalVOC.default_resolution == [384, 384]
    assert D.KITTI.default_resolution == [384, 1280]
    for cls in D.dataset_factory.values():
        assert cls.mean.shape == cls.std.shape == (1, 1, 3) and cls.mean.dtype == np.float32
        assert len(cls.class_name) == cls.num_classes + 1 and cls.class_name[0] == "__background__"
    assert np.allclose(D.COCO.mean.reshape(-1), [0.40789654, 0.44719302, 0.47026115], atol=0)
    assert len(D.COCO._valid_ids) == 80 and D.COCO.cat_ids[90] == 79 and D.COCOHP.flip_idx[0] == [1, 2]
    assert D.KITTI.class_name[1:] == ["Pedestrian", "Car", "Cyclist"] and D.PascalVOC.class_name[15] == "person"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
@pytest.mark.parametrize("name,file,cls", [("coco", "coco.py", "COCO"), ("coco_hp", "coco_hp.py", "COCOHP"),
                                           ("pascal", "pascal.py", "PascalVOC"), ("kitti", "kitti.py", "KITTI")])
def test_class_attributes_equal_the_reference(name, file, cls):
    stub = types.ModuleType("pycocotools")
    stub.coco = types.ModuleType("pycocotools.coco")
    stub.cocoeval = types.ModuleType("pycocotools.cocoeval")
    stub.cocoeval.COCOeval = object
    saved = {k: sys.modules.get(k) for k in ("pycocotools", "pycocotools.coco", "pycocotools.cocoeval", "cv2")}
    sys.modules.update({"pycocotools": stub, "pycocotools.coco": stub.coco, "pycocotools.cocoeval": stub.cocoeval,
                        "cv2": types.ModuleType("cv2")})
    try:
        spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, file))
        mod = importlib.util.module_from_spec(spec)
        sys.dont_write_bytecode = True
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    ref, mine = getattr(mod, cls), D.dataset_factory[name]
    for attr in ("num_classes", "default_resolution"):
        assert getattr(ref, attr) == getattr(mine, attr), attr
    for attr in ("mean", "std"):
        a, b = getattr(ref, attr), getattr(mine, attr)
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), attr
    if name == "coco_hp":
        assert ref.flip_idx == mine.flip_idx and ref.num_joints == mine.num_joints
    # instance attributes of the reference (class names, id maps) by reading its constructor's source
    src = open(os.path.join(REF, file)).read()
    for cname in ([] if name == "coco_hp" else mine.class_name[1:]):      # (coco_hp.py names no classes: one, the person)
        assert ("'%s'" % cname) in src or ('"%s"' % cname) in src, cname


@pytest.mark.parametrize("task,dataset", [("ctdet", "coco"), ("ctdet", "pascal"), ("multi_pose", "coco_hp"),
                                          ("ddd", "kitti"), ("exdet", "coco")])
def test_update_dataset_info_takes_a_descriptor(task, dataset):
    """test.py:50-56: ``opt = opts().update_dataset_info_and_set_heads(opt, Dataset)``."""
    o = opts()
    opt = o.parse([task, "--dataset", dataset])
    Dataset = D.dataset_factory[opt.dataset]
    opt = o.update_dataset_info_and_set_heads(opt, Dataset)
    assert opt.num_classes == Dataset.num_classes and (opt.input_h, opt.input_w) == tuple(Dataset.default_resolution)
    assert opt.mean is Dataset.mean and (opt.output_h, opt.output_w) == (opt.input_h // 4, opt.input_w // 4)
    if (task, dataset) == ("ctdet", "pascal"):
        assert opt.heads == {"hm": 20, "wh": 2, "reg": 2}


def test_writers_equal_the_reference_goldens(tmp_path):
    gen = _gen()
    golden = json.load(open(os.path.join(HERE, "golden", "results_golden.json")))
    for task, cls in (("ctdet", D.COCO), ("multi_pose", D.COCOHP)):
        ds = cls(None, "val")
        assert ds.convert_eval_format(gen.results_inputs(task)) == golden[task]
        ds.save_results(gen.results_inputs(task), str(tmp_path))
        assert json.load(open(tmp_path / "results.json")) == golden[task]
    inp = gen.pascal_inputs()
    voc = D.PascalVOC(None, "val", images=sorted(inp))
    assert len(voc) == 4 and voc.convert_eval_format(inp) == golden["pascal"]
    voc.save_results(inp, str(tmp_path))
    assert json.load(open(tmp_path / "results.json")) == golden["pascal"]
    got = voc.convert_eval_format(inp)
    assert len(got) == 21 and got[0] == [[], [], [], []] and all(len(c) == 4 for c in got)
    with pytest.raises(NotImplementedError):
        voc.run_eval(inp, str(tmp_path))
    # KITTI: label files, through the descriptor
    from test_tasks_host import GEN as TG, GOLD as TGOLD
    kd = D.KITTI(None, "val")
    assert kd.convert_eval_format({}) is None
    kd.save_results(TG.kitti_results_inputs(TGOLD), str(tmp_path))
    want = json.load(open(os.path.join(HERE, "golden", "tasks_kitti_golden.json")))
    assert {n: open(tmp_path / "results" / n).read() for n in os.listdir(tmp_path / "results")} == want
