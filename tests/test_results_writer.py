"""COCO result writer vs goldens produced by the reference's own convert_eval_format
(tests/golden/gen_golden_results.py).  CPU only; exact equality (every float goes through
"{:.2f}".format, so JSON round trips are exact)."""
import copy
import importlib.util
import json
import os

import numpy as np
import pytest

from centernet_amd import results as R

HERE = os.path.dirname(os.path.abspath(__file__))


def _gen():
    spec = importlib.util.spec_from_file_location(
        "gen_golden_results", os.path.join(HERE, "golden", "gen_golden_results.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("task", ["ctdet", "multi_pose"])
def test_convert_eval_format_matches_reference(task, tmp_path):
    golden = json.load(open(os.path.join(HERE, "golden", "results_golden.json")))[task]
    inp = _gen().results_inputs(task)
    keep = copy.deepcopy(inp)
    got = R.convert_eval_format(inp, task)
    assert got == golden
    # inputs untouched (the reference's in-place xyxy->xywh is not reproduced)
    for i in inp:
        for c in inp[i]:
            assert np.array_equal(np.asarray(inp[i][c]), np.asarray(keep[i][c]))
    R.save_results(inp, str(tmp_path), task)
    assert json.load(open(tmp_path / "results.json")) == golden


def test_valid_ids_are_coco_categories():
    assert len(R.COCO_VALID_IDS) == 80 and R.COCO_VALID_IDS[0] == 1 and R.COCO_VALID_IDS[-1] == 90
    assert R.COCO_VALID_IDS == sorted(set(R.COCO_VALID_IDS))
