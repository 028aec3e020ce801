"""Host tail of the path (mirror of src/lib/utils/post_process.py:83-114): map
detections from the output grid back to image coordinates and split per class."""
import numpy as np

from .image import transform_preds


def _split_by_class(rows, classes, num_classes):
    """{1-based class: rows of that class in their original order} via one stable sort."""
    order = np.argsort(classes, kind='stable')
    rows, classes = rows[order], classes[order]
    edges = np.searchsorted(classes, np.arange(num_classes + 1))
    return {j + 1: rows[edges[j]:edges[j + 1]].tolist() for j in range(num_classes)}


def ctdet_post_process(dets, c, s, h, w, num_classes):
    """(B, K, 6) detections in output-grid units -> per image ``{class: [[x1, y1, x2, y2,
    score], ...]}`` in source-frame pixels, classes 1-based (what utils/post_process.py:83-100
    returns).  Both box corners go through one inverse map; rows are grouped by class with a
    stable sort instead of ``num_classes`` boolean masks.  ``dets`` is not modified."""
    ret = []
    for i in range(dets.shape[0]):
        corners = transform_preds(dets[i, :, 0:4].reshape(-1, 2), c[i], s[i], (w, h))
        rows = np.concatenate([corners.reshape(-1, 4).astype(np.float32),
                               dets[i, :, 4:5].astype(np.float32)], axis=1)
        ret.append(_split_by_class(rows, dets[i, :, -1].astype(np.int64), num_classes))
    return ret


def multi_pose_post_process(dets, c, s, h, w):
    """(B, K, 40) pose detections -> per image ``{1: [[x1, y1, x2, y2, score, 17 x (x, y)],
    ...]}`` in source-frame pixels (utils/post_process.py:103-114): the 2 box corners and the 17
    joints of every detection are one (K * 19, 2) point set under one inverse map."""
    ret = []
    for i in range(dets.shape[0]):
        K = dets.shape[1]
        pts = np.concatenate([dets[i, :, 0:4].reshape(K, 2, 2), dets[i, :, 5:39].reshape(K, 17, 2)],
                             axis=1)
        moved = transform_preds(pts.reshape(-1, 2), c[i], s[i], (w, h)).reshape(K, 19, 2)
        rows = np.concatenate([moved[:, :2].reshape(K, 4), dets[i, :, 4:5],
                               moved[:, 2:].reshape(K, 34)], axis=1).astype(np.float32)
        ret.append({1: rows.tolist()})
    return ret


def ctdet_results_batch(dets, metas, num_classes, scale=1, max_per_image=100):
    """Vectorised host tail for a batch, single scale, no NMS: for every image exactly what
    ``merge_outputs([post_process(dets[i], meta_i, scale)])`` returns (detectors/ctdet.py:47-73)
    -- same float64 affine, same float32 rounding, same per-class row order -- without the
    80-class Python loop per image.  Images that share their geometry (every frame of a video:
    same centre, extent and output grid) share ONE inverse map and go through it together; the
    class grouping of the whole batch is one stable sort."""
    from .image import apply_affine, get_affine_transform
    dets = np.asarray(dets)
    B, K, _ = dets.shape
    # inverse maps: one per distinct (centre, extent, output grid)
    groups = {}
    for i, m in enumerate(metas):
        key = (np.asarray(m['c'], np.float32).tobytes(), np.asarray(m['s'], np.float32).tobytes(),
               int(m['out_width']), int(m['out_height']))
        groups.setdefault(key, []).append(i)
    xy = np.empty((B, K, 4), np.float32)
    for idx in groups.values():
        m = metas[idx[0]]
        to_source = get_affine_transform(m['c'], m['s'], 0, (m['out_width'], m['out_height']), inv=1)
        sel = idx if len(idx) < B else slice(None)
        pts = dets[sel, :, 0:4].reshape(-1, 2)
        xy[sel] = apply_affine(pts, to_source).astype(np.float32).reshape(-1, K, 4)
    rows = np.concatenate([xy, dets[:, :, 4:5].astype(np.float32)], axis=2)
    rows[:, :, :4] /= scale
    cls = dets[:, :, 5].astype(np.int64)
    stray = bool(((cls < 0) | (cls >= num_classes)).any())
    if K > max_per_image or stray:
        # keep the max_per_image best of every image (ties at the threshold kept, as the
        # reference's np.partition test does): rare, handled image by image.  The same path
        # serves class ids outside [0, num_classes) (a head with more channels than
        # opt.num_classes): such rows match no `classes == j` of the reference
        # (post_process.py:93-99) and are dropped, never moved into another image's rows.
        out = []
        for i in range(B):
            keep = np.ones(K, bool)
            if K > max_per_image:
                kth = K - max_per_image
                thresh = np.partition(rows[i, :, 4], kth)[kth]
                keep = rows[i, :, 4] >= thresh
            r, c = rows[i][keep], cls[i][keep]
            order = np.argsort(c, kind='stable')
            r, c = r[order], c[order]
            bounds = np.searchsorted(c, np.arange(num_classes + 1))
            out.append({j + 1: r[bounds[j]:bounds[j + 1]] for j in range(num_classes)})
        return out
    order = np.argsort(cls, axis=1, kind='stable')
    rows = np.take_along_axis(rows, order[:, :, None], axis=1)
    cls = np.take_along_axis(cls, order, axis=1)
    # class boundaries of every image at once: position of (image, class) in the flattened keys
    flat = (cls + np.arange(B, dtype=np.int64)[:, None] * num_classes).reshape(-1)
    bounds = np.searchsorted(flat, np.arange(B * num_classes + 1)).tolist()
    rows = rows.reshape(B * K, 5)
    out = []
    for i in range(B):
        b0 = i * num_classes
        out.append({j + 1: rows[bounds[b0 + j]:bounds[b0 + j + 1]] for j in range(num_classes)})
    return out
