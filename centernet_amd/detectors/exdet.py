"""exdet task -- ExtremeNet-style detection (public behaviour of src/lib/detectors/exdet.py:23-118):
four extreme-point heat-maps + a centre map per class on the HIP network, grouped into boxes by
``cn_exct_decode_f32`` (K^4 candidate scoring with the edge aggregation of --aggr_weight in front);
--agnostic_ex: one extreme-point map per edge for all classes, ``cn_agnex_ct_decode_f32``.

Two things the reference's class does are kept as they are, because they ARE its results:
* ``post_process`` reads the decode's rows as TWO images -- the frame and its mirror image -- and
  un-mirrors the second half's box (exdet.py:87-91): the task is meant to run with --flip_test (its
  batch is then [frame, mirrored frame]); without it the second half of the single image's rows is
  mirrored all the same;
* only the box corners go through the inverse affine; the eight extreme-point coordinates of a row
  stay in output-grid units (exdet.py:92-96) and are dropped by ``merge_outputs``.
One thing differs: the reference's ``merge_outputs`` calls ``soft_nms`` without importing it
(exdet.py:110 -- a NameError as shipped); here it is the library's (external/nms.pyx:77-170)."""
import time

import numpy as np
import torch

from ..decode import agnex_ct_decode, exct_decode
from ..image import transform_preds
from ..soft_nms import soft_nms
from .base_detector import BaseDetector

EDGE_MAPS = ('hm_t', 'hm_l', 'hm_b', 'hm_r', 'hm_c')
EDGE_OFFSETS = ('reg_t', 'reg_l', 'reg_b', 'reg_r')


class ExdetDetector(BaseDetector):
    def __init__(self, opt):
        if opt.K > 64:
            # the reference would score K^4 = 10^8 groupings per image at the --K default of 100; the
            # kernel takes K <= 64 (cn_exct_decode_f32), ExtremeNet's own setting is 40
            raise ValueError("the exdet task needs --K <= 64 (K^4 candidate groupings per image); use --K 40")
        super(ExdetDetector, self).__init__(opt)
        self.decode = agnex_ct_decode if opt.agnostic_ex else exct_decode      # exdet.py:26

    def process(self, images, return_time=False):
        """exdet.py:28-55: the five maps post-sigmoid (in place, as there), then ``exct_decode`` with
        the sub-pixel offsets of the four edges when the network has them -> (B, 1000, 14)."""
        with torch.no_grad():
            output = self.model(images, borrow=True, check=True)[-1]
            heats = [output[n].sigmoid_() for n in EDGE_MAPS]
            torch.cuda.synchronize()
            forward_time = time.time()
            offsets = [output[n] for n in EDGE_OFFSETS] if self.opt.reg_offset else []
            dets = self.decode(*(heats + offsets), K=self.opt.K, scores_thresh=self.opt.scores_thresh,
                               center_thresh=self.opt.center_thresh, aggr_weight=self.opt.aggr_weight)
        return (output, dets, forward_time) if return_time else (output, dets)

    def post_process(self, dets, meta, scale=1):
        """(B, 1000, 14) rows [x1, y1, x2, y2, score, 8 extreme-point coords, class] -> ONE (n, 14)
        array: second half un-mirrored, box corners in source pixels of the unscaled frame."""
        out_w, out_h = meta['out_width'], meta['out_height']
        rows = dets.detach().cpu().numpy().reshape(2, -1, 14)
        left, right = rows[1, :, 0].copy(), rows[1, :, 2].copy()
        rows[1, :, 0], rows[1, :, 2] = out_w - right, out_w - left
        rows = rows.reshape(1, -1, 14)
        rows[0, :, 0:2] = transform_preds(rows[0, :, 0:2], meta['c'], meta['s'], (out_w, out_h))
        rows[0, :, 2:4] = transform_preds(rows[0, :, 2:4], meta['c'], meta['s'], (out_w, out_h))
        rows[:, :, 0:4] /= scale
        return rows[0]

    def merge_outputs(self, detections):
        """All scales together, rows with a positive score only, soft-NMS (Gaussian, Nt 0.5) per class,
        then the ``max_per_image`` best over all classes, ties kept (exdet.py:99-123)."""
        rows = np.concatenate(list(detections), axis=0).astype(np.float32)
        rows = rows[rows[:, 4] > 0]
        classes = rows[:, -1]
        results = {}
        for j in range(self.num_classes):
            boxes = np.ascontiguousarray(rows[classes == j][:, 0:5])    # (the routine touches columns 0-4 only)
            soft_nms(boxes, Nt=0.5, method=2)
            results[j + 1] = boxes
        scores = np.hstack([results[j][:, -1] for j in range(1, self.num_classes + 1)])
        if len(scores) > self.max_per_image:
            kth = len(scores) - self.max_per_image
            thresh = np.partition(scores, kth)[kth]
            for j in range(1, self.num_classes + 1):
                results[j] = results[j][results[j][:, -1] >= thresh]
        return results

    def _pipe_for(self, frames, depth):
        raise NotImplementedError("run_frames / run_frames_stream: not built for the exdet task; use run(frame)")
