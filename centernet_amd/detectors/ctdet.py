"""CtdetDetector (mirror of src/lib/detectors/ctdet.py:23-73)."""
import time

import numpy as np
import torch

from ..decode import ctdet_decode
from ..post_process import ctdet_post_process, ctdet_results_batch
from ..utils import flip_tensor
from .base_detector import BaseDetector


class CtdetDetector(BaseDetector):
    def __init__(self, opt):
        super(CtdetDetector, self).__init__(opt)

    def process(self, images, return_time=False):
        # ctdet.py:28-45.  hm.sigmoid_() is fused into the decode kernel unless the
        # flip-test average (which needs the sigmoid values) is requested.
        with torch.no_grad():
            output = self.model(images)[-1]
            hm = output['hm']
            wh = output['wh']
            reg = output['reg'] if self.opt.reg_offset else None
            fused_sigmoid = not self.opt.flip_test
            if self.opt.flip_test:
                hm = hm.sigmoid_()
                hm = (hm[0:1] + flip_tensor(hm[1:2])) / 2
                wh = (wh[0:1] + flip_tensor(wh[1:2])) / 2
                reg = reg[0:1] if reg is not None else None
            torch.cuda.synchronize()
            forward_time = time.time()
            dets = ctdet_decode(hm, wh, reg=reg, cat_spec_wh=self.opt.cat_spec_wh, K=self.opt.K,
                                apply_sigmoid=fused_sigmoid)
        if return_time:
            return output, dets, forward_time
        return output, dets

    def post_process(self, dets, meta, scale=1):
        # ctdet.py:47-56
        dets = dets.detach().cpu().numpy()
        dets = dets.reshape(1, -1, dets.shape[2])
        dets = ctdet_post_process(dets.copy(), [meta['c']], [meta['s']], meta['out_height'],
                                  meta['out_width'], self.opt.num_classes)
        for j in range(1, self.num_classes + 1):
            dets[0][j] = np.array(dets[0][j], dtype=np.float32).reshape(-1, 5)
            dets[0][j][:, :4] /= scale
        return dets[0]

    def merge_outputs(self, detections):
        # ctdet.py:58-73
        results = {}
        for j in range(1, self.num_classes + 1):
            results[j] = np.concatenate([d[j] for d in detections], axis=0).astype(np.float32)
            if len(self.scales) > 1 or self.opt.nms:
                from ..soft_nms import soft_nms
                soft_nms(results[j], Nt=0.5, method=2)
        scores = np.hstack([results[j][:, 4] for j in range(1, self.num_classes + 1)])
        if len(scores) > self.max_per_image:
            kth = len(scores) - self.max_per_image
            thresh = np.partition(scores, kth)[kth]
            for j in range(1, self.num_classes + 1):
                keep_inds = (results[j][:, 4] >= thresh)
                results[j] = results[j][keep_inds]
        return results

    def run_batch(self, images):
        """NEW surface: ``images`` (B,3,H,W) fp32 already normalised, on the device.
        Returns the raw (B,K,6) detections in output-grid units (device tensor)."""
        with torch.no_grad():
            output = self.model(images)[-1]
            return ctdet_decode(output['hm'], output['wh'],
                                reg=output['reg'] if self.opt.reg_offset else None,
                                cat_spec_wh=self.opt.cat_spec_wh, K=self.opt.K, apply_sigmoid=True)

    def run_frames(self, frames):
        """NEW surface (the reference's test loop is batch_size=1, test.py:60-62): a list of
        (H, W, 3) uint8 BGR frames of one size -> list of per-image result dicts, exactly what
        ``run(frame)['results']`` returns for each (single scale, no flip).  The frames are
        uploaded as uint8, pre-processed on the device straight into one batch tensor, and the
        whole batch goes through the network + decode once."""
        assert len(self.scales) == 1 and not self.opt.flip_test, "run_frames is single-scale, no flip"
        scale = self.scales[0]
        batch, metas = None, []
        if len({tuple(f.shape) for f in frames}) != 1:
            raise ValueError("run_frames needs frames of one size")
        stacked = torch.from_numpy(np.ascontiguousarray(np.stack(frames))).to(self.opt.device)
        for i, f in enumerate(stacked):
            if batch is None:
                probe, meta = self.pre_process_device(f, scale)
                batch = torch.empty((len(frames),) + tuple(probe.shape[1:]), device=probe.device,
                                    dtype=torch.float32)
                batch[0:1].copy_(probe)
            else:
                _, meta = self.pre_process_device(f, scale, out=batch[i:i + 1])
            metas.append(meta)
        dets = self.run_batch(batch).detach().cpu().numpy()
        return ctdet_results_batch(dets, metas, self.opt.num_classes, scale, self.max_per_image)
