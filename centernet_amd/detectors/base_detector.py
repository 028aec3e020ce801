"""BaseDetector (mirror of src/lib/detectors/base_detector.py:16-143).

Same public surface: ``pre_process``, ``process``, ``post_process``, ``merge_outputs``,
``run(image_or_path_or_tensor, meta=None)`` returning ``{'results', 'tot', 'load',
'pre', 'net', 'dec', 'post', 'merge'}``.  New surface (the reference is single-image
only): ``run_batch(images)`` for device-resident batches.
"""
import time

import numpy as np
import torch

from ..image import get_affine_transform, warp_affine, resize_bilinear
from ..model import create_model, load_model
from ..native import NativeError


class BaseDetector(object):
    def __init__(self, opt):
        if opt.gpus[0] >= 0:
            opt.device = torch.device('cuda')
        else:
            raise NativeError("--gpus -1 (CPU) is not supported: centernet_amd is the MI355X "
                              "path and has no CPU fallback")
        print('Creating model...')
        self.model = create_model(opt.arch, opt.heads, opt.head_conv)
        if opt.load_model:
            self.model = load_model(self.model, opt.load_model)
        self.model = self.model.to(opt.device)
        self.model.eval()
        self.mean = np.array(opt.mean, dtype=np.float32).reshape(1, 1, 3)
        self.std = np.array(opt.std, dtype=np.float32).reshape(1, 1, 3)
        self.max_per_image = 100
        self.num_classes = opt.num_classes
        self.scales = opt.test_scales
        self.opt = opt
        self.pause = True

    def pre_process(self, image, scale, meta=None):
        # base_detector.py:37-65
        height, width = image.shape[0:2]
        new_height = int(height * scale)
        new_width = int(width * scale)
        if self.opt.fix_res:
            inp_height, inp_width = self.opt.input_h, self.opt.input_w
            c = np.array([new_width / 2., new_height / 2.], dtype=np.float32)
            s = max(height, width) * 1.0
        else:
            inp_height = (new_height | self.opt.pad) + 1
            inp_width = (new_width | self.opt.pad) + 1
            c = np.array([new_width // 2, new_height // 2], dtype=np.float32)
            s = np.array([inp_width, inp_height], dtype=np.float32)
        trans_input = get_affine_transform(c, s, 0, [inp_width, inp_height])
        resized_image = resize_bilinear(image, (new_width, new_height))
        inp_image = warp_affine(resized_image, trans_input, (inp_width, inp_height))
        inp_image = ((inp_image / 255. - self.mean) / self.std).astype(np.float32)
        images = inp_image.transpose(2, 0, 1).reshape(1, 3, inp_height, inp_width)
        if self.opt.flip_test:
            images = np.concatenate((images, images[:, :, :, ::-1]), axis=0)
        images = torch.from_numpy(np.ascontiguousarray(images))
        meta = {'c': c, 's': s, 'out_height': inp_height // self.opt.down_ratio,
                'out_width': inp_width // self.opt.down_ratio}
        return images, meta

    def pre_process_device(self, image, scale, meta=None, out=None):
        """pre_process with the resize / warp / normalise / CHW (/ flip) steps on the device
        (cn_resize_bilinear_u8 + cn_warp_normalize_u8_f32): the uint8 frame is uploaded and the
        fp32 (1|2,3,H,W) batch is produced in HBM.  Same arithmetic as ``pre_process``
        (bit-identical output); used by ``run`` for ndarray / path inputs on a HIP device."""
        import ctypes
        from .. import native
        from ..image import invert_affine
        lib = native.lib()
        height, width = image.shape[0:2]
        new_height = int(height * scale)
        new_width = int(width * scale)
        if self.opt.fix_res:
            inp_height, inp_width = self.opt.input_h, self.opt.input_w
            c = np.array([new_width / 2., new_height / 2.], dtype=np.float32)
            s = max(height, width) * 1.0
        else:
            inp_height = (new_height | self.opt.pad) + 1
            inp_width = (new_width | self.opt.pad) + 1
            c = np.array([new_width // 2, new_height // 2], dtype=np.float32)
            s = np.array([inp_width, inp_height], dtype=np.float32)
        trans_input = get_affine_transform(c, s, 0, [inp_width, inp_height])
        dev = self.opt.device
        if torch.is_tensor(image):   # already uploaded (run_frames: one H2D copy per batch)
            if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3 or \
                    not image.is_cuda or not image.is_contiguous():
                raise ValueError("pre_process_device needs a contiguous (H, W, 3) uint8 HIP tensor")
            src = image
        else:
            if image.dtype != np.uint8 or image.ndim != 3 or image.shape[2] != 3:
                raise ValueError("pre_process_device needs an (H, W, 3) uint8 BGR image")
            src = torch.from_numpy(np.ascontiguousarray(image)).to(dev)
        st = native.stream_ptr()
        if (new_height, new_width) != (height, width):
            resized = torch.empty((new_height, new_width, 3), device=dev, dtype=torch.uint8)
            native.check(lib.cn_resize_bilinear_u8(native.ptr(src), height, width, width * 3,
                                                   new_height, new_width, native.ptr(resized), st),
                         "cn_resize_bilinear_u8")
            src = resized
        nb = 2 if self.opt.flip_test else 1
        if out is not None:   # caller-provided slice of a batch tensor (run_frames)
            assert tuple(out.shape) == (nb, 3, inp_height, inp_width) and out.is_contiguous()
            images = out
        else:
            images = torch.empty((nb, 3, inp_height, inp_width), device=dev, dtype=torch.float32)
        mi = (ctypes.c_double * 6)(*invert_affine(trans_input).reshape(-1))
        mean = (ctypes.c_float * 3)(*[float(v) for v in self.mean.reshape(-1)])
        std = (ctypes.c_float * 3)(*[float(v) for v in self.std.reshape(-1)])
        native.check(lib.cn_warp_normalize_u8_f32(native.ptr(src), new_height, new_width,
                                                  new_width * 3, mi, inp_height, inp_width, mean,
                                                  std, int(self.opt.flip_test), native.ptr(images),
                                                  st),
                     "cn_warp_normalize_u8_f32")
        meta = {'c': c, 's': s, 'out_height': inp_height // self.opt.down_ratio,
                'out_width': inp_width // self.opt.down_ratio}
        return images, meta

    def process(self, images, return_time=False):
        raise NotImplementedError

    def post_process(self, dets, meta, scale=1):
        raise NotImplementedError

    def merge_outputs(self, detections):
        raise NotImplementedError

    def debug(self, debugger, images, dets, output, scale=1):
        raise NotImplementedError("visual debugging (cv2/matplotlib) is outside the hot path")

    def show_results(self, debugger, image, results):
        raise NotImplementedError("visual debugging (cv2/matplotlib) is outside the hot path")

    def _load_image(self, path):
        from PIL import Image  # cv2.imread replacement: BGR uint8
        rgb = np.asarray(Image.open(path).convert('RGB'))
        return np.ascontiguousarray(rgb[:, :, ::-1])

    def run(self, image_or_path_or_tensor, meta=None):
        # base_detector.py:82-143 (no Debugger construction: debug==0 path only)
        load_time, pre_time, net_time, dec_time, post_time = 0, 0, 0, 0, 0
        merge_time, tot_time = 0, 0
        start_time = time.time()
        pre_processed = False
        if isinstance(image_or_path_or_tensor, np.ndarray):
            image = image_or_path_or_tensor
        elif type(image_or_path_or_tensor) == type(''):
            image = self._load_image(image_or_path_or_tensor)
        else:
            image = image_or_path_or_tensor['image'][0].numpy()
            pre_processed_images = image_or_path_or_tensor
            pre_processed = True
        loaded_time = time.time()
        load_time += (loaded_time - start_time)
        detections = []
        for scale in self.scales:
            scale_start_time = time.time()
            if not pre_processed:
                on_device = getattr(self.opt.device, 'type', str(self.opt.device)) == 'cuda'
                if on_device and not getattr(self.opt, 'host_pre_process', False) and \
                        image.dtype == np.uint8:
                    images, meta = self.pre_process_device(image, scale, meta)
                else:
                    images, meta = self.pre_process(image, scale, meta)
            else:
                images = pre_processed_images['images'][scale][0]
                meta = pre_processed_images['meta'][scale]
                meta = {k: v.numpy()[0] for k, v in meta.items()}
            images = images.to(self.opt.device)
            torch.cuda.synchronize()
            pre_process_time = time.time()
            pre_time += pre_process_time - scale_start_time
            output, dets, forward_time = self.process(images, return_time=True)
            torch.cuda.synchronize()
            net_time += forward_time - pre_process_time
            decode_time = time.time()
            dec_time += decode_time - forward_time
            if self.opt.debug >= 2:
                self.debug(None, images, dets, output, scale)
            dets = self.post_process(dets, meta, scale)
            torch.cuda.synchronize()
            post_process_time = time.time()
            post_time += post_process_time - decode_time
            detections.append(dets)
        results = self.merge_outputs(detections)
        torch.cuda.synchronize()
        end_time = time.time()
        merge_time += end_time - post_process_time
        tot_time += end_time - start_time
        if self.opt.debug >= 1:
            self.show_results(None, image, results)
        return {'results': results, 'tot': tot_time, 'load': load_time, 'pre': pre_time,
                'net': net_time, 'dec': dec_time, 'post': post_time, 'merge': merge_time}
