"""Dataset descriptors -- what the inference side of the reference reads off its dataset classes
(``dataset_factory[opt.dataset]``, src/lib/datasets/dataset_factory.py:15-20; used by test.py:50-56,
89-95 as ``opts().update_dataset_info_and_set_heads(opt, Dataset)`` and, after the run, as
``dataset.run_eval(results, save_dir)``): class count, default input resolution, per-channel mean / std
(the dataset classes carry more digits than the ``opts().init()`` defaults: 0.40789654 vs 0.408), class
names, category-id maps, the left / right joint pairs, and the result writers.

Only that: no annotation loading, no samplers, no augmentation, no evaluators -- ``run_eval`` needs
pycocotools / the KITTI and VOC evaluation tools and stays with the reference's classes.  A descriptor is
constructed without files; ``images`` (the evaluation set's image ids, in order) is needed by the Pascal
writer only, whose result file is indexed by image position."""
import json

import numpy as np

from . import results as _results


def _rgb_stats(values):
    return np.array(values, dtype=np.float32).reshape(1, 1, 3)


class _Dataset(object):
    def __init__(self, opt=None, split='val', images=None):
        self.opt, self.split = opt, split
        self.images = list(images) if images is not None else []
        self.num_samples = len(self.images)

    def __len__(self):
        return self.num_samples

    def _to_float(self, x):
        return float("{:.2f}".format(x))

    def run_eval(self, results, save_dir):
        raise NotImplementedError("evaluation needs the benchmark's own tools (pycocotools, the KITTI / VOC "
                                  "evaluators): write the result file with save_results and run them on it")


class COCO(_Dataset):
    """datasets/dataset/coco.py:13-64,86-112 (ctdet, exdet)."""
    num_classes = 80
    default_resolution = [512, 512]
    mean = _rgb_stats([0.40789654, 0.44719302, 0.47026115])
    std = _rgb_stats([0.28863828, 0.27408164, 0.27809835])
    class_name = ['__background__'] + (
        "person|bicycle|car|motorcycle|airplane|bus|train|truck|boat|traffic light|fire hydrant|stop sign|"
        "parking meter|bench|bird|cat|dog|horse|sheep|cow|elephant|bear|zebra|giraffe|backpack|umbrella|"
        "handbag|tie|suitcase|frisbee|skis|snowboard|sports ball|kite|baseball bat|baseball glove|skateboard|"
        "surfboard|tennis racket|bottle|wine glass|cup|fork|knife|spoon|bowl|banana|apple|sandwich|orange|"
        "broccoli|carrot|hot dog|pizza|donut|cake|chair|couch|potted plant|bed|dining table|toilet|tv|laptop|"
        "mouse|remote|keyboard|cell phone|microwave|oven|toaster|sink|refrigerator|book|clock|vase|scissors|"
        "teddy bear|hair drier|toothbrush").split("|")
    _valid_ids = list(_results.COCO_VALID_IDS)
    cat_ids = {v: i for i, v in enumerate(_valid_ids)}
    max_objs = 128
    task = "ctdet"

    def convert_eval_format(self, all_bboxes):
        return _results.convert_eval_format(all_bboxes, self.task, self._valid_ids)

    def save_results(self, results, save_dir):
        _results.save_results(results, save_dir, self.task, self._valid_ids)


class COCOHP(COCO):
    """datasets/dataset/coco_hp.py:13-103 (multi_pose): one class, 17 joints."""
    num_classes = 1
    num_joints = 17
    flip_idx = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
    edges = [[0, 1], [0, 2], [1, 3], [2, 4], [4, 6], [3, 5], [5, 6], [5, 7], [7, 9], [6, 8], [8, 10], [6, 12],
             [5, 11], [11, 12], [12, 14], [14, 16], [11, 13], [13, 15]]
    class_name = ['__background__', 'person']      # (not named by the reference's class: category 1 of its json)
    _valid_ids = [1]
    cat_ids = {1: 0}
    max_objs = 32
    task = "multi_pose"


class PascalVOC(_Dataset):
    """datasets/dataset/pascal.py:13-80 (ctdet on VOC 07+12)."""
    num_classes = 20
    default_resolution = [384, 384]
    mean = _rgb_stats([0.485, 0.456, 0.406])
    std = _rgb_stats([0.229, 0.224, 0.225])
    class_name = ['__background__'] + ("aeroplane|bicycle|bird|boat|bottle|bus|car|cat|chair|cow|diningtable|dog|"
                                       "horse|motorbike|person|pottedplant|sheep|sofa|train|tvmonitor").split("|")
    _valid_ids = np.arange(1, 21, dtype=np.int32)
    cat_ids = {int(v): i for i, v in enumerate(_valid_ids)}
    max_objs = 50

    def convert_eval_format(self, all_bboxes):
        """``detections[class][image position]`` = that image's rows of the class as lists, class 0 (the
        background slot) left empty (pascal.py:55-65) -- the layout tools/reval.py reads."""
        detections = [[[] for _ in range(self.num_samples)] for _ in range(self.num_classes + 1)]
        for pos, img_id in enumerate(self.images):
            for j in range(1, self.num_classes + 1):
                rows = all_bboxes[img_id][j]
                detections[j][pos] = rows.tolist() if isinstance(rows, np.ndarray) else rows
        return detections

    def save_results(self, results, save_dir):
        with open('{}/results.json'.format(save_dir), 'w') as f:
            json.dump(self.convert_eval_format(results), f)


class KITTI(_Dataset):
    """datasets/dataset/kitti.py:17-82 (ddd)."""
    num_classes = 3
    default_resolution = [384, 1280]
    mean = _rgb_stats([0.485, 0.456, 0.406])
    std = _rgb_stats([0.229, 0.224, 0.225])
    class_name = list(_results.KITTI_CLASS_NAMES)
    cat_ids = {1: 0, 2: 1, 3: 2, 4: -3, 5: -3, 6: -2, 7: -99, 8: -99, 9: -1}
    max_objs = 50

    def convert_eval_format(self, all_bboxes):
        pass        # as the reference: the label files are the format (kitti.py:65-66)

    def save_results(self, results, save_dir):
        _results.save_results_kitti(results, save_dir, self.class_name)


dataset_factory = {'coco': COCO, 'pascal': PascalVOC, 'kitti': KITTI, 'coco_hp': COCOHP}
