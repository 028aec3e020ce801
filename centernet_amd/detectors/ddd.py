"""ddd task -- monocular 3-D detection (public behaviour of src/lib/detectors/ddd.py:22-91): centre
heat-map + depth + orientation bins + box dimensions (+ 2-D size and sub-pixel offset) on the HIP
network, decoded by ``cn_ddd_decode_f32``, lifted to camera coordinates on the host
(``post_process.ddd_post_process``).  ``run(image, calib)``: the second argument is the frame's
3 x 4 projection matrix (test.py:37-39,105-106); without it the detector's KITTI default is used."""
import time

import numpy as np
import torch

from ..decode import ddd_decode
from ..image import get_affine_transform, warp_affine
from ..post_process import ddd_post_process
from .base_detector import BaseDetector


class DddDetector(BaseDetector):
    def __init__(self, opt):
        super(DddDetector, self).__init__(opt)
        self.calib = np.array([[707.0493, 0, 604.0814, 45.75831],
                               [0, 707.0493, 180.5066, -0.3454157],
                               [0, 0, 1., 0.004981016]], dtype=np.float32)     # ddd.py:25-27

    # ------------------------------------------------------------------ pre-process
    def _frame_geometry(self, height, width):
        """Centre, extent and the frame -> network-input map (ddd.py:31-42): the frame is NOT resized
        or padded to a multiple; it is warped straight onto the fixed input size, the extent being
        the frame's own (width, height) -- or the input's under --keep_res -- as int32, x first."""
        inp_h, inp_w = self.opt.input_h, self.opt.input_w
        c = np.array([width / 2, height / 2], dtype=np.float32)
        s = np.array([inp_w, inp_h] if self.opt.keep_res else [width, height], dtype=np.int32)
        return c, s, get_affine_transform(c, s, 0, [inp_w, inp_h])

    def _meta(self, c, s, calib):
        return {'c': c, 's': s, 'out_height': self.opt.input_h // self.opt.down_ratio,
                'out_width': self.opt.input_w // self.opt.down_ratio,
                'calib': self.calib if calib is None else np.array(calib, dtype=np.float32)}

    def pre_process(self, image, scale, calib=None):
        """Host form (ddd.py:30-54); ``scale`` is accepted and unused, as in the reference.  The
        normalisation is the ddd class's own FLOAT32 chain ``(u8 / 255 - mean) / std`` (ddd.py:45-46) --
        not the float64-then-round of the other tasks (base_detector.py:56) -- taken through a 256-entry
        table per channel built with exactly those float32 operations."""
        c, s, to_input = self._frame_geometry(image.shape[0], image.shape[1])
        warped = warp_affine(image, to_input, (self.opt.input_w, self.opt.input_h))
        levels = np.arange(256, dtype=np.float32).reshape(256, 1, 1) / 255.
        table = ((levels - self.mean) / self.std).reshape(256, 3)             # float32 throughout
        batch = np.stack([table[warped[:, :, ch], ch] for ch in range(3)])[None]
        return torch.from_numpy(np.ascontiguousarray(batch)), self._meta(c, s, calib)

    def pre_process_device(self, image, scale, calib=None, out=None):
        """The device pre-process kernels implement the float64 normalisation of the other tasks; the
        ddd chain differs from it in the last bit of some levels, so this task warps and normalises on
        the host (one frame per call, as the reference) and uploads the float32 batch."""
        if torch.is_tensor(image):
            raise ValueError("the ddd task pre-processes host frames: (H, W, 3) uint8 BGR arrays")
        images, meta = self.pre_process(image, scale, calib)
        images = images.to(self.opt.device)
        if out is not None:
            out.copy_(images)
            images = out
        return images, meta

    # ------------------------------------------------------------------ network + decode
    def process(self, images, return_time=False):
        """ddd.py:56-73: post-sigmoid centre map, depth = 1 / (sigmoid(dep) + 1e-6) - 1, then
        ``ddd_decode``.  The returned ``output`` holds the maps in that transformed state."""
        with torch.no_grad():
            output = self.model(images, borrow=True, check=True)[-1]
            output['hm'] = output['hm'].sigmoid_()
            output['dep'] = 1. / (output['dep'].sigmoid() + 1e-6) - 1.
            wh = output['wh'] if self.opt.reg_bbox else None
            reg = output['reg'] if self.opt.reg_offset else None
            torch.cuda.synchronize()
            forward_time = time.time()
            dets = ddd_decode(output['hm'], output['rot'], output['dep'], output['dim'], wh=wh, reg=reg,
                              K=self.opt.K)
        return (output, dets, forward_time) if return_time else (output, dets)

    def post_process(self, dets, meta, scale=1):
        """(1, K, 18) rows in output-grid units -> {class: (n, 13) float32 [alpha, x1, y1, x2, y2, h, w,
        l, x, y, z, rotation_y, score]} in source pixels / camera metres (ddd.py:75-80)."""
        host = dets.detach().cpu().numpy()
        per_image = ddd_post_process(host.copy(), [meta['c']], [meta['s']], [meta['calib']], self.opt)
        self.this_calib = meta['calib']
        return per_image[0]

    def merge_outputs(self, detections):
        """Single scale: the first entry, every class cut at --peak_thresh (ddd.py:82-88)."""
        results = detections[0]
        for j in range(1, self.num_classes + 1):
            if len(results[j]) > 0:
                results[j] = results[j][results[j][:, -1] > self.opt.peak_thresh]
        return results

    def _pipe_for(self, frames, depth):
        raise NotImplementedError("run_frames / run_frames_stream: the ddd task has a per-frame calibration "
                                  "matrix and a host pre-process; use run(frame, calib) or run_batch")

    def run_batch(self, images, probe=None):
        """New surface (as CtdetDetector.run_batch): a device-resident, normalised batch -> raw
        (B, K, 18) rows of ``ddd_decode``; the centre map's sigmoid is fused into the decode."""
        self._note_unchecked_forward()
        with torch.no_grad():
            o = self.model(images, borrow=True)[-1]
            dep = 1. / (o['dep'].sigmoid() + 1e-6) - 1.
            return ddd_decode(o['hm'], o['rot'], dep, o['dim'], wh=o['wh'] if self.opt.reg_bbox else None,
                              reg=o['reg'] if self.opt.reg_offset else None, K=self.opt.K,
                              apply_sigmoid=True)
