"""ctdet task (public behaviour of src/lib/detectors/ctdet.py:23-73): centre heat-map, box size
and sub-pixel offset decoded by the fused ``cn_ctdet_decode_f32`` kernels.  New surface:
``run_batch`` (device-resident batches) and ``run_frames`` (lists of uint8 frames)."""
import time

import numpy as np
import torch

from ..decode import ctdet_decode
from ..post_process import ctdet_results_batch
from ..utils import flip_average
from .base_detector import BaseDetector


class CtdetDetector(BaseDetector):
    def __init__(self, opt):
        super(CtdetDetector, self).__init__(opt)

    def _decode(self, hm, wh, reg, logits):
        return ctdet_decode(hm, wh, reg=reg, cat_spec_wh=self.opt.cat_spec_wh, K=self.opt.K,
                            apply_sigmoid=logits)

    def process(self, images, return_time=False):
        """Network + decode of one pre-processed batch (ctdet.py:28-45).  Without flip-test the
        sigmoid of ctdet.py:31 is fused into the decode kernel (``hm`` stays logits); with it,
        the mirrored frame (image 1) is averaged in after the sigmoid, as in the reference."""
        with torch.no_grad():
            # consumed before the next run; check=True: f32s range words read here (the
            # reference synchronises at this point too), re-calibrate + re-run on a clamped value
            output = self.model(images, borrow=True, check=True)[-1]
            hm, wh = output['hm'], output['wh']
            reg = output['reg'] if self.opt.reg_offset else None
            logits = not self.opt.flip_test
            if self.opt.flip_test:
                hm = flip_average(hm, sigmoid=True)      # (sigmoid_() of both images in place, as ctdet.py:31)
                wh = flip_average(wh)
                reg = None if reg is None else reg[0:1]
            torch.cuda.synchronize()
            forward_time = time.time()
            dets = self._decode(hm, wh, reg, logits)
        return (output, dets, forward_time) if return_time else (output, dets)

    def post_process(self, dets, meta, scale=1):
        """(1, K, 6) in output-grid units -> {class: (n, 5) float32} in the coordinates of the
        unscaled frame (ctdet.py:47-56)."""
        host = dets.detach().cpu().numpy()
        host = host.reshape(1, -1, host.shape[2])
        # the batch tail with a batch of one and no cap: same rows, same order, same float32
        # rounding as ctdet_post_process + the per-class np.array / "/= scale" loop, without
        # the trip through Python lists
        return ctdet_results_batch(host, [meta], self.opt.num_classes, scale,
                                   max_per_image=host.shape[1])[0]

    def merge_outputs(self, detections):
        """Concatenate the test scales per class, soft-NMS when asked or when there are several,
        then keep the ``max_per_image`` best over all classes by score threshold -- ``>=``, so
        ties may exceed the cap, as in the reference (ctdet.py:58-73)."""
        classes = range(1, self.num_classes + 1)
        if len(detections) == 1 and not (len(self.scales) > 1 or self.opt.nms):
            results = {c: detections[0][c] for c in classes}     # nothing to merge, nothing edited
        else:
            results = {c: np.concatenate([d[c] for d in detections], axis=0).astype(np.float32)
                       for c in classes}
        if len(self.scales) > 1 or self.opt.nms:
            from ..soft_nms import soft_nms
            for c in classes:
                soft_nms(results[c], Nt=0.5, method=2)
        scores = np.hstack([results[c][:, 4] for c in classes])
        if len(scores) > self.max_per_image:
            kth = len(scores) - self.max_per_image
            thresh = np.partition(scores, kth)[kth]
            for c in classes:
                results[c] = results[c][results[c][:, 4] >= thresh]
        return results

    # ------------------------------------------------------------------ new surface
    def run_batch(self, images, probe=None):
        """``images`` (B,3,H,W) fp32, already normalised, on the device -> raw (B,K,6)
        detections in output-grid units (device tensor).  Asynchronous: nothing here waits for
        the device, so the f32s range words of the forward are NOT looked at yet -- call
        ``range_ok()`` where the results are consumed (``run_frames`` does; a pipeline checks
        once per synchronisation point, the words accumulate over the forwards in between).
        ``probe``: optional dict for measurement (bench.py): ``event_after`` (set of launch
        indices) in, ``net_events`` (HIP events at those launch boundaries) and ``dec_events``
        (before / after the decode) out."""
        self._note_unchecked_forward()
        with torch.no_grad():
            if probe is None:
                out = self.model(images, borrow=True)[-1]
                return self._decode(out['hm'], out['wh'],
                                    out['reg'] if self.opt.reg_offset else None, True)
            probe['net_events'] = []
            out = self.model(images, borrow=True, events=probe['net_events'],
                             event_after=probe.get('event_after'))[-1]
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            dets = self._decode(out['hm'], out['wh'], out['reg'] if self.opt.reg_offset else None,
                                True)
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            probe['dec_events'] = (e0, e1)
            return dets

    # ---- device tail of the frame pipeline (base_detector._FramePipe)
    def _device_tail_alloc(self, pipe):
        """Buffers of cn_ctdet_post_process_f32 for one pipe, or None when the host tail has to
        serve it (more detections than the kernel takes or than max_per_image keeps)."""
        from ..image import get_affine_transform
        K, nc, B, dev = self.opt.K, self.opt.num_classes, pipe.B, self.opt.device
        if K > 128 or K > self.max_per_image:
            return None
        m = pipe.meta
        to_source = get_affine_transform(m['c'], m['s'], 0, (m['out_width'], m['out_height']), inv=1)
        t = {'to_source': torch.from_numpy(np.ascontiguousarray(to_source, np.float64).reshape(-1)).to(dev),
             'rows': torch.empty((B, K, 5), device=dev, dtype=torch.float32),
             'bounds': torch.empty((B, nc + 1), device=dev, dtype=torch.int32),
             'rows_host': [torch.empty((B, K, 5), dtype=torch.float32).pin_memory() for _ in range(pipe.depth)],
             'bounds_host': [torch.empty((B, nc + 1), dtype=torch.int32).pin_memory() for _ in range(pipe.depth)]}
        return t

    def _device_tail_run(self, pipe, slot, dets):
        from .. import native
        t, K, nc = pipe.tail, self.opt.K, self.opt.num_classes
        dets = dets.contiguous()
        native.check(native.lib().cn_ctdet_post_process_f32(
            native.ptr(dets), pipe.B, K, nc, native.ptr(t['to_source']), 0, float(pipe.scale),
            native.ptr(t['rows']), native.ptr(t['bounds']), native.stream_ptr()), "cn_ctdet_post_process_f32")
        t['rows_host'][slot].copy_(t['rows'], non_blocking=True)
        t['bounds_host'][slot].copy_(t['bounds'], non_blocking=True)

    def _device_tail_results(self, pipe, slot, n):
        """Per image ``{class: (n, 5) float32}`` -- the rows are already in source pixels and grouped
        by class; what is left is 80 slices per image."""
        t, nc = pipe.tail, self.opt.num_classes
        rows = t['rows_host'][slot].numpy().copy()        # (the pinned buffer is reused by a later batch)
        bounds = t['bounds_host'][slot].numpy().tolist()
        out = []
        for i in range(n):
            r, bd = rows[i], bounds[i]
            out.append({j + 1: r[bd[j]:bd[j + 1]] for j in range(nc)})
        return out

    def results_batch(self, dets, metas, scale):
        """Host tail of ``run_frames``: (B, K, 6) host array -> per-image ``{class: (n, 5)}``."""
        return ctdet_results_batch(dets, metas, self.opt.num_classes, scale, self.max_per_image)
