"""multi_pose task (public behaviour of src/lib/detectors/multi_pose.py:24-81): centre
heat-map + box size + 17 joint offsets, optional joint heat-maps / sub-pixel offsets, decoded
by the fused ``cn_multi_pose_decode_f32`` kernels."""
import time

import numpy as np
import torch

from ..decode import multi_pose_decode
from ..post_process import multi_pose_post_process
from ..utils import flip_average
from .base_detector import BaseDetector

ROW = 39  # [x1, y1, x2, y2, score, 17 x (x, y)]


class MultiPoseDetector(BaseDetector):
    def __init__(self, opt):
        super(MultiPoseDetector, self).__init__(opt)
        self.flip_idx = opt.flip_idx

    def _head_maps(self, out):
        """Post-sigmoid centre map and the optional branches, as the decode expects them."""
        hm = out['hm'].sigmoid_()
        hm_hp = None
        if self.opt.hm_hp:
            hm_hp = out['hm_hp'] if self.opt.mse_loss else out['hm_hp'].sigmoid_()
        reg = out['reg'] if self.opt.reg_offset else None
        hp_offset = out['hp_offset'] if self.opt.reg_hp_offset else None
        return hm, out['wh'], out['hps'], reg, hm_hp, hp_offset

    def _average_flip(self, hm, wh, hps, reg, hm_hp, hp_offset):
        """Flip-test (multi_pose.py:44-55): image 1 of the batch is the mirrored frame; maps are
        averaged after un-mirroring, offsets of the un-mirrored frame are kept."""
        hm = flip_average(hm)
        wh = flip_average(wh)
        hps = flip_average(hps, self.flip_idx, offsets=True)
        if hm_hp is not None:
            hm_hp = flip_average(hm_hp, self.flip_idx)
        reg = None if reg is None else reg[0:1]
        hp_offset = None if hp_offset is None else hp_offset[0:1]
        return hm, wh, hps, reg, hm_hp, hp_offset

    def process(self, images, return_time=False):
        with torch.no_grad():
            torch.cuda.synchronize()
            # consumed before the next run; f32s range words checked here (see CtdetDetector)
            output = self.model(images, borrow=True, check=True)[-1]
            maps = self._head_maps(output)
            torch.cuda.synchronize()
            forward_time = time.time()
            if self.opt.flip_test:
                maps = self._average_flip(*maps)
                output['hm'], output['wh'], output['hps'] = maps[0], maps[1], maps[2]
            hm, wh, hps, reg, hm_hp, hp_offset = maps
            dets = multi_pose_decode(hm, wh, hps, reg=reg, hm_hp=hm_hp, hp_offset=hp_offset,
                                     K=self.opt.K)
        return (output, dets, forward_time) if return_time else (output, dets)

    def run_batch(self, images, probe=None):
        """New surface (as CtdetDetector.run_batch): a device-resident, normalised batch ->
        raw (B,K,40) detections in output-grid units; sigmoids fused into the decode kernels."""
        self._note_unchecked_forward()
        with torch.no_grad():
            ev = None
            if probe is not None:
                ev = probe['net_events'] = []
            o = self.model(images, borrow=True, events=ev,
                           event_after=None if probe is None else probe.get('event_after'))[-1]
            if probe is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            if self.opt.mse_loss and self.opt.hm_hp:
                raise NotImplementedError("run_batch: mse_loss joint heat-maps are not logits")
            dets = multi_pose_decode(o['hm'], o['wh'], o['hps'],
                                     reg=o['reg'] if self.opt.reg_offset else None,
                                     hm_hp=o['hm_hp'] if self.opt.hm_hp else None,
                                     hp_offset=o['hp_offset'] if self.opt.reg_hp_offset else None,
                                     K=self.opt.K, apply_sigmoid=True)
            if probe is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                probe['dec_events'] = (e0, e1)
            return dets

    def results_batch(self, dets, metas, scale):
        """Host tail of ``run_frames``: (B, K, 40) host array -> per image ``{1: [[x1, y1, x2,
        y2, score, 17 x (x, y)], ...]}``, i.e. merge_outputs([post_process(...)]) for one scale
        without NMS (multi_pose.py:62-81)."""
        per = multi_pose_post_process(dets.copy(), [m['c'] for m in metas], [m['s'] for m in metas],
                                      metas[0]['out_height'], metas[0]['out_width'])
        out = []
        for d in per:
            rows = np.array(d[1], dtype=np.float32).reshape(-1, ROW)
            rows[:, :4] /= scale
            rows[:, 5:] /= scale
            out.append({1: rows.tolist()})
        return out

    def post_process(self, dets, meta, scale=1):
        """Output-grid units -> image coordinates of the unscaled frame (multi_pose.py:62-72)."""
        host = dets.detach().cpu().numpy()
        host = host.reshape(1, -1, host.shape[2])
        per_class = multi_pose_post_process(host.copy(), [meta['c']], [meta['s']],
                                            meta['out_height'], meta['out_width'])[0]
        for cls in range(1, self.num_classes + 1):
            rows = np.array(per_class[cls], dtype=np.float32).reshape(-1, ROW)
            rows[:, :4] /= scale
            rows[:, 5:] /= scale
            per_class[cls] = rows
        return per_class

    def merge_outputs(self, detections):
        """Concatenate the test scales; soft-NMS when asked or when there are several
        (multi_pose.py:74-81).  The person class is the only one."""
        people = np.concatenate([d[1] for d in detections], axis=0).astype(np.float32)
        if self.opt.nms or len(self.opt.test_scales) > 1:
            from ..soft_nms import soft_nms_39
            soft_nms_39(people, Nt=0.5, method=2)
        return {1: people.tolist()}
