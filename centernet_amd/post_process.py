"""Host tail of the path (mirror of src/lib/utils/post_process.py:83-114, and :10-81 for the ddd
task): map detections from the output grid back to image coordinates and split per class."""
import numpy as np

from .ddd_utils import ddd2locrot
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


def get_alpha(rot):
    """(n, 8) orientation head [bin 1: 2 class logits, sin, cos | bin 2: the same] -> observation
    angle (utils/post_process.py:13-21): the bin whose second logit is larger wins; bin 1 is centred
    on -pi / 2, bin 2 on +pi / 2.  The blend is written with the reference's 0 / 1 products (a float64
    result, as there)."""
    first = rot[:, 1] > rot[:, 5]
    a1 = np.arctan2(rot[:, 2], rot[:, 3]) + (-0.5 * np.pi)
    a2 = np.arctan2(rot[:, 6], rot[:, 7]) + (0.5 * np.pi)
    return a1 * first + a2 * (1 - first)


def ddd_post_process_2d(dets, c, s, opt):
    """(B, K, 16 | 18) rows of ``ddd_decode`` [x, y, score, rot 8, depth, dim 3, (w, h), class] ->
    per image {1-based class: (n, 8 | 10) float32 [x, y, score, alpha, depth, h, w, l, (w, h)]} with
    the centre in source pixels (utils/post_process.py:24-51).  The reference sends the (w, h) pair
    through the same point map as the centre -- translation included; so does this.  Writes the
    moved centres into ``dets`` (callers pass a copy, as the reference's detector does)."""
    out = []
    has_wh = dets.shape[2] > 16
    grid = (opt.output_w, opt.output_h)
    for i in range(dets.shape[0]):
        dets[i, :, :2] = transform_preds(dets[i, :, 0:2], c[i], s[i], grid)
        cls = dets[i, :, -1]
        per_class = {}
        for j in range(opt.num_classes):
            rows = dets[i, cls == j]
            cols = [rows[:, :3].astype(np.float32),
                    get_alpha(rows[:, 3:11])[:, np.newaxis].astype(np.float32),
                    rows[:, 11:12].astype(np.float32), rows[:, 12:15].astype(np.float32)]
            if has_wh:
                cols.append(transform_preds(rows[:, 15:17], c[i], s[i], grid).astype(np.float32))
            per_class[j + 1] = np.concatenate(cols, axis=1)
        out.append(per_class)
    return out


def ddd_post_process_3d(dets, calibs):
    """The 2-D stage's rows -> per image {class: (n, 13) float32 [alpha, x1, y1, x2, y2, h, w, l,
    x, y, z, rotation_y, score]}, an EMPTY class being a (0,) array (utils/post_process.py:53-79).
    Every image is lifted with ``calibs[0]``, as in the reference (its detector is single-image)."""
    out = []
    for per_class in dets:
        lifted = {}
        for cls, rows in per_class.items():
            preds = []
            for r in rows:
                center, score, alpha, depth, dims, wh = r[:2], r[2], r[3], r[4], r[5:8], r[8:10]
                location, rotation_y = ddd2locrot(center, alpha, dims, depth, calibs[0])
                box = [center[0] - wh[0] / 2, center[1] - wh[1] / 2,
                       center[0] + wh[0] / 2, center[1] + wh[1] / 2]
                preds.append([alpha] + box + dims.tolist() + location.tolist() + [rotation_y, score])
            lifted[cls] = np.array(preds, dtype=np.float32)
        out.append(lifted)
    return out


def ddd_post_process(dets, c, s, calibs, opt):
    """utils/post_process.py:81-86."""
    return ddd_post_process_3d(ddd_post_process_2d(dets, c, s, opt), calibs)


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
