"""Result writers -- the on-disk formats the benchmarks' evaluators read: COCO json for
ctdet / exdet / multi_pose, KITTI label files for ddd.

Mirror of ``COCO.convert_eval_format / save_results`` (src/lib/datasets/dataset/coco.py:86-112)
and ``COCOHP.convert_eval_format / save_results`` (src/lib/datasets/dataset/coco_hp.py:70-103):
xyxy -> xywh, every float rounded through ``"{:.2f}".format`` (:86-87), contiguous class index ->
COCO category id through ``_valid_ids`` (:52-60).  ``run_eval`` needs pycocotools and the
annotation files and stays with the reference's dataset classes.

One deliberate difference: the reference subtracts in place (``bbox[2] -= bbox[0]``, coco.py:96-97)
and so corrupts the caller's result arrays; here the inputs are left untouched.
"""
import json

import numpy as np

COCO_VALID_IDS = [
    1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 27, 28,
    31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55,
    56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 67, 70, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 84,
    85, 86, 87, 88, 89, 90]


def _to_float(x):
    return float("{:.2f}".format(x))


def convert_eval_format(all_bboxes, task="ctdet", valid_ids=None):
    """``all_bboxes``: {image_id: {class (1-based): rows}} as returned by ``run()['results']``.
    ctdet rows: [x1,y1,x2,y2,score(,8 extreme-point coords)]; multi_pose rows:
    [x1,y1,x2,y2,score,34 keypoint coords]."""
    valid_ids = COCO_VALID_IDS if valid_ids is None else valid_ids
    detections = []
    for image_id in all_bboxes:
        for cls_ind in all_bboxes[image_id]:
            category_id = 1 if task == "multi_pose" else valid_ids[cls_ind - 1]
            for row in all_bboxes[image_id][cls_ind]:
                bbox = [row[0], row[1], row[2] - row[0], row[3] - row[1]]
                det = {"image_id": int(image_id), "category_id": int(category_id),
                       "bbox": [_to_float(v) for v in bbox], "score": _to_float(row[4])}
                if task == "multi_pose":
                    kps = np.concatenate([np.array(row[5:39], dtype=np.float32).reshape(-1, 2),
                                          np.ones((17, 1), dtype=np.float32)], axis=1)
                    det["keypoints"] = [_to_float(v) for v in kps.reshape(51).tolist()]
                elif len(row) > 5:
                    det["extreme_points"] = [_to_float(v) for v in row[5:13]]
                detections.append(det)
    return detections


def save_results(results, save_dir, task="ctdet", valid_ids=None):
    with open('{}/results.json'.format(save_dir), 'w') as f:
        json.dump(convert_eval_format(results, task, valid_ids), f)


KITTI_CLASS_NAMES = ['__background__', 'Pedestrian', 'Car', 'Cyclist']      # datasets/dataset/kitti.py:35-36


def kitti_result_lines(per_class, class_names=None):
    """One image's ddd results ``{class (1-based): (n, 13) rows [alpha, x1, y1, x2, y2, h, w, l, x, y, z,
    rotation_y, score]}`` -> the lines of its KITTI label file: ``<type> 0.0 0`` (truncation, occlusion) and
    the 13 values with two decimals (datasets/dataset/kitti.py:74-81)."""
    names = KITTI_CLASS_NAMES if class_names is None else class_names
    lines = []
    for cls_ind in per_class:
        for row in per_class[cls_ind]:
            lines.append('{} 0.0 0'.format(names[cls_ind]) + ''.join(' {:.2f}'.format(v) for v in row))
    return lines


def save_results_kitti(results, save_dir, class_names=None):
    """``{image id: run()['results']}`` -> ``<save_dir>/results/<image id, six digits>.txt`` for the KITTI
    object evaluator (datasets/dataset/kitti.py:68-82)."""
    import os
    results_dir = os.path.join(save_dir, 'results')
    os.makedirs(results_dir, exist_ok=True)
    for img_id in results:
        with open(os.path.join(results_dir, '{:06d}.txt'.format(img_id)), 'w') as f:
            for line in kitti_result_lines(results[img_id], class_names):
                f.write(line + '\n')
