"""oracle/post_oracle.py -- TEST INFRASTRUCTURE (see oracle/__init__.py).

NumPy restatement of the host tail of the path, following the reference statement by
statement (including its per-point Python loop):
  transform_preds / get_affine_transform / affine_transform   utils/image.py:19-66
  ctdet_post_process                                           utils/post_process.py:83-100
  CtdetDetector.post_process / merge_outputs                   detectors/ctdet.py:47-73
cv2.getAffineTransform (third-party, absent here) is restated from OpenCV's published source
(pre_oracle.cv_get_affine_transform: its 6x6 system, LU with partial pivoting in double); the
closed-form cases in tests/test_host.py / tests/test_oracle_pre.py pin it (no reference test
covers it, no OpenCV build is available offline).
"""
import numpy as np


def _get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]


def _third(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def _cv_get_affine(src, dst):
    # cv::getAffineTransform: the interleaved 6x6 system + LU with partial pivoting, restated once
    # in oracle/pre_oracle.py (cv_get_affine_transform)
    from .pre_oracle import cv_get_affine_transform
    return cv_get_affine_transform(src, dst)


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32),
                         inv=0):
    if not isinstance(scale, np.ndarray) and not isinstance(scale, list):
        scale = np.array([scale, scale], dtype=np.float32)
    src_w = scale[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    src_dir = _get_dir([0, src_w * -0.5], rot_rad)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale * shift
    src[1, :] = center + src_dir + scale * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    src[2:, :] = _third(src[0, :], src[1, :])
    dst[2:, :] = _third(dst[0, :], dst[1, :])
    return _cv_get_affine(np.float32(dst), np.float32(src)) if inv else \
        _cv_get_affine(np.float32(src), np.float32(dst))


def affine_transform(pt, t):
    new_pt = np.array([pt[0], pt[1], 1.], dtype=np.float32).T
    return np.dot(t, new_pt)[:2]


def transform_preds(coords, center, scale, output_size):
    target = np.zeros(coords.shape)
    trans = get_affine_transform(center, scale, 0, output_size, inv=1)
    for p in range(coords.shape[0]):
        target[p, 0:2] = affine_transform(coords[p, 0:2], trans)
    return target


def ctdet_post_process(dets, c, s, h, w, num_classes):
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


def ctdet_results(dets, meta, num_classes, scale=1, max_per_image=100):
    """detectors/ctdet.py:47-73 for one image, single scale, no NMS."""
    d = dets.reshape(1, -1, dets.shape[2]).copy()
    d = ctdet_post_process(d, [meta['c']], [meta['s']], meta['out_height'], meta['out_width'],
                           num_classes)[0]
    for j in range(1, num_classes + 1):
        d[j] = np.array(d[j], dtype=np.float32).reshape(-1, 5)
        d[j][:, :4] /= scale
    scores = np.hstack([d[j][:, 4] for j in range(1, num_classes + 1)])
    if len(scores) > max_per_image:
        kth = len(scores) - max_per_image
        thresh = np.partition(scores, kth)[kth]
        for j in range(1, num_classes + 1):
            d[j] = d[j][d[j][:, 4] >= thresh]
    return d


def soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """external/nms.pyx:77-170 (and :172-275 for 39-column rows), statement by statement;
    ``boxes`` float32 (N, 5|39) modified in place; returns the kept indices.  The reference's
    pinned bit-for-bit (kept count and the whole in-place array, rows past N included) against
    the reference's own cython build, oracle/_ref (tests/test_oracle_ref.py)."""
    f = np.float32
    N = boxes.shape[0]
    ncol = boxes.shape[1]
    for i in range(boxes.shape[0]):
        if i >= N:
            break
        maxscore = boxes[i, 4]
        maxpos = i
        tmp = boxes[i].copy()
        pos = i + 1
        while pos < N:
            if maxscore < boxes[pos, 4]:
                maxscore = boxes[pos, 4]
                maxpos = pos
            pos += 1
        boxes[i, :ncol] = boxes[maxpos, :ncol]
        boxes[maxpos, :ncol] = tmp
        tx1, ty1, tx2, ty2 = boxes[i, 0], boxes[i, 1], boxes[i, 2], boxes[i, 3]
        pos = i + 1
        while pos < N:
            x1, y1, x2, y2 = boxes[pos, 0], boxes[pos, 1], boxes[pos, 2], boxes[pos, 3]
            # cython emits the integer literals as the double 1.0: float differences widen to
            # double, every assignment to a `cdef float` rounds once (generated C of nms.pyx:133-143)
            d = np.float64
            area = f((d(f(x2 - x1)) + 1.0) * (d(f(y2 - y1)) + 1.0))
            iw = f(d(f(min(tx2, x2) - max(tx1, x1))) + 1.0)
            if iw > 0:
                ih = f(d(f(min(ty2, y2) - max(ty1, y1))) + 1.0)
                if ih > 0:
                    ua = f(((d(f(tx2 - tx1)) + 1.0) * (d(f(ty2 - ty1)) + 1.0) + d(area)) - d(f(iw * ih)))
                    ov = f(f(iw * ih) / ua)
                    if method == 1:
                        weight = f(1.0 - d(ov)) if ov > Nt else f(1)
                    elif method == 2:
                        weight = f(np.exp(float(f(-f(ov * ov) / f(sigma)))))
                    else:
                        weight = f(0) if ov > Nt else f(1)
                    boxes[pos, 4] = f(weight * boxes[pos, 4])
                    if boxes[pos, 4] < threshold:
                        boxes[pos, :5] = boxes[N - 1, :5]
                        if ncol > 5:                      # nms.pyx:260-268: columns 5.. swap
                            t = boxes[pos, 5:].copy()
                            boxes[pos, 5:] = boxes[N - 1, 5:]
                            boxes[N - 1, 5:] = t
                        N -= 1
                        pos -= 1
            pos += 1
    return list(range(N))


def multi_pose_post_process(dets, c, s, h, w):
    """utils/post_process.py:103-114, statement by statement."""
    ret = []
    for i in range(dets.shape[0]):
        bbox = transform_preds(dets[i, :, :4].reshape(-1, 2), c[i], s[i], (w, h))
        pts = transform_preds(dets[i, :, 5:39].reshape(-1, 2), c[i], s[i], (w, h))
        top_preds = np.concatenate([bbox.reshape(-1, 4), dets[i, :, 4:5], pts.reshape(-1, 34)],
                                   axis=1).astype(np.float32).tolist()
        ret.append({np.ones(1, dtype=np.int32)[0]: top_preds})
    return ret


def multi_pose_results(dets, meta, scale=1):
    """detectors/multi_pose.py:62-81 for one image, single scale, no NMS."""
    d = dets.reshape(1, -1, dets.shape[2])
    d = multi_pose_post_process(d.copy(), [meta['c']], [meta['s']], meta['out_height'],
                                meta['out_width'])
    rows = np.array(d[0][1], dtype=np.float32).reshape(-1, 39)
    rows[:, :4] /= scale
    rows[:, 5:] /= scale
    return {1: rows.tolist()}


def ctdet_post_process_scale(dets, meta, num_classes, scale=1):
    """CtdetDetector.post_process, detectors/ctdet.py:47-56."""
    d = dets.reshape(1, -1, dets.shape[2])
    d = ctdet_post_process(d.copy(), [meta['c']], [meta['s']], meta['out_height'],
                           meta['out_width'], num_classes)
    for j in range(1, num_classes + 1):
        d[0][j] = np.array(d[0][j], dtype=np.float32).reshape(-1, 5)
        d[0][j][:, :4] /= scale
    return d[0]


def ctdet_merge_outputs(detections, num_classes, n_scales, nms=False, max_per_image=100):
    """CtdetDetector.merge_outputs, detectors/ctdet.py:58-73 (the list soft_nms returns is
    discarded there: only its in-place score decay and row swaps matter)."""
    results = {}
    for j in range(1, num_classes + 1):
        results[j] = np.concatenate([d[j] for d in detections], axis=0).astype(np.float32)
        if n_scales > 1 or nms:
            soft_nms(results[j], Nt=0.5, method=2)
    scores = np.hstack([results[j][:, 4] for j in range(1, num_classes + 1)])
    if len(scores) > max_per_image:
        kth = len(scores) - max_per_image
        thresh = np.partition(scores, kth)[kth]
        for j in range(1, num_classes + 1):
            keep_inds = (results[j][:, 4] >= thresh)
            results[j] = results[j][keep_inds]
    return results


# ---------------------------------------------------------------------------------------------
# ddd task: utils/post_process.py:10-86 + utils/ddd_utils.py:68-114, detectors/ddd.py:75-88
# ---------------------------------------------------------------------------------------------
def ddd_alpha(rot):
    """get_alpha, utils/post_process.py:13-21."""
    idx = rot[:, 1] > rot[:, 5]
    alpha1 = np.arctan2(rot[:, 2], rot[:, 3]) + (-0.5 * np.pi)
    alpha2 = np.arctan2(rot[:, 6], rot[:, 7]) + (0.5 * np.pi)
    return alpha1 * idx + alpha2 * (1 - idx)


def ddd_unproject(pt_2d, depth, P):
    """unproject_2d_to_3d, utils/ddd_utils.py:68-78."""
    z = depth - P[2, 3]
    x = (pt_2d[0] * depth - P[0, 3] - P[0, 2] * z) / P[0, 0]
    y = (pt_2d[1] * depth - P[1, 3] - P[1, 2] * z) / P[1, 1]
    return np.array([x, y, z], dtype=np.float32)


def ddd_rot_y(alpha, x, cx, fx):
    """alpha2rot_y, utils/ddd_utils.py:80-92."""
    rot_y = alpha + np.arctan2(x - cx, fx)
    if rot_y > np.pi:
        rot_y -= 2 * np.pi
    if rot_y < -np.pi:
        rot_y += 2 * np.pi
    return rot_y


def ddd_results(dets, meta, num_classes, out_w, out_h):
    """DddDetector.post_process (detectors/ddd.py:75-80) for one image: ddd_post_process_2d then
    ddd_post_process_3d, statement by statement.  dets (1, K, 18) as ddd_decode returns them."""
    d = dets.reshape(1, -1, dets.shape[2]).copy()
    c, s, calib = meta['c'], meta['s'], meta['calib']
    d[0, :, :2] = transform_preds(d[0, :, 0:2], c, s, (out_w, out_h))
    classes = d[0, :, -1]
    out = {}
    for j in range(num_classes):
        inds = (classes == j)
        rows = np.concatenate([
            d[0, inds, :3].astype(np.float32), ddd_alpha(d[0, inds, 3:11])[:, np.newaxis].astype(np.float32),
            d[0, inds, 11:12].astype(np.float32), d[0, inds, 12:15].astype(np.float32),
            transform_preds(d[0, inds, 15:17], c, s, (out_w, out_h)).astype(np.float32)], axis=1)
        preds = []
        for r in rows:
            center, score, alpha, depth, dimensions, wh = r[:2], r[2], r[3], r[4], r[5:8], r[8:10]
            locations = ddd_unproject(center, depth, calib)                       # ddd2locrot, ddd_utils.py:109-114
            locations[1] += dimensions[0] / 2
            rotation_y = ddd_rot_y(alpha, center[0], calib[0, 2], calib[0, 0])
            bbox = [center[0] - wh[0] / 2, center[1] - wh[1] / 2, center[0] + wh[0] / 2, center[1] + wh[1] / 2]
            preds.append([alpha] + bbox + dimensions.tolist() + locations.tolist() + [rotation_y, score])
        out[j + 1] = np.array(preds, dtype=np.float32)
    return out


def ddd_merge_outputs(detections, num_classes, peak_thresh):
    """DddDetector.merge_outputs, detectors/ddd.py:82-88."""
    results = detections[0]
    for j in range(1, num_classes + 1):
        if len(results[j]) > 0:
            results[j] = results[j][results[j][:, -1] > peak_thresh]
    return results


# ---------------------------------------------------------------------------------------------
# exdet task: detectors/exdet.py:86-123
# ---------------------------------------------------------------------------------------------
def exdet_post_process(dets, meta, scale=1):
    """ExdetDetector.post_process, detectors/exdet.py:86-97: the rows of [frame, mirrored frame]."""
    out_width, out_height = meta['out_width'], meta['out_height']
    d = np.array(dets, dtype=np.float32).reshape(2, -1, 14)
    d[1, :, [0, 2]] = out_width - d[1, :, [2, 0]]
    d = d.reshape(1, -1, 14)
    d[0, :, 0:2] = transform_preds(d[0, :, 0:2], meta['c'], meta['s'], (out_width, out_height))
    d[0, :, 2:4] = transform_preds(d[0, :, 2:4], meta['c'], meta['s'], (out_width, out_height))
    d[:, :, 0:4] /= scale
    return d[0]


def exdet_merge_outputs(detections, num_classes, max_per_image=100):
    """ExdetDetector.merge_outputs, detectors/exdet.py:99-123 (soft_nms: this file's restatement of
    external/nms.pyx, pinned to the cython build in tests/test_oracle_ref.py)."""
    detections = np.concatenate([d for d in detections], axis=0).astype(np.float32)
    classes = detections[..., -1]
    keep = detections[:, 4] > 0
    detections, classes = detections[keep], classes[keep]
    results = {}
    for j in range(num_classes):
        results[j + 1] = detections[classes == j][:, 0:7].astype(np.float32)
        soft_nms(results[j + 1], Nt=0.5, method=2)
        results[j + 1] = results[j + 1][:, 0:5]
    scores = np.hstack([results[j][:, -1] for j in range(1, num_classes + 1)])
    if len(scores) > max_per_image:
        kth = len(scores) - max_per_image
        thresh = np.partition(scores, kth)[kth]
        for j in range(1, num_classes + 1):
            results[j] = results[j][results[j][:, -1] >= thresh]
    return results
