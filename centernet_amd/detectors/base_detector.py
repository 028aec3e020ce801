"""Task API base class.

Public surface and semantics follow the reference's ``BaseDetector``
(src/lib/detectors/base_detector.py:16-143): ``pre_process``, ``process``, ``post_process``,
``merge_outputs`` and ``run(image_or_path_or_tensor, meta=None)`` which returns
``{'results', 'tot', 'load', 'pre', 'net', 'dec', 'post', 'merge'}``.  The implementation is
organised differently: the input geometry lives in one helper shared by the host and the device
pre-process, the time buckets are kept by a small phase clock, and ``run`` sends uint8 frames
through the device pre-process (``cn_resize_bilinear_u8`` + ``cn_warp_normalize_u8_f32``).
New surface (the reference is single-image only): ``run_batch`` / ``run_frames`` in the task
classes.
"""
import collections
import ctypes
import time

import numpy as np
import torch

from .. import native
from ..image import get_affine_transform, invert_affine, normalize_chw, resize_bilinear, warp_affine
from ..model import create_model, load_model

InputGeometry = collections.namedtuple(
    "InputGeometry", "src_h src_w scaled_h scaled_w inp_h inp_w center extent")


class _PhaseClock(object):
    """Wall-clock buckets of ``run`` ('load', 'pre', 'net', 'dec', 'post', 'merge', 'tot').
    A lap synchronises the device first -- the reference's explicit
    ``torch.cuda.synchronize()`` calls (base_detector.py:112-134)."""

    def __init__(self):
        self.t = dict.fromkeys(("load", "pre", "net", "dec", "post", "merge", "tot"), 0.0)
        self._start = self._last = time.time()

    def lap(self, bucket, sync=True, until=None):
        if sync:
            torch.cuda.synchronize()
        now = time.time() if until is None else until
        self.t[bucket] += now - self._last
        self._last = now

    def finish(self):
        self.t["tot"] = self._last - self._start
        return self.t


class _FramePipe(object):
    """Persistent resources of ``run_frames`` / ``run_frames_stream`` for one batch geometry
    (B frames of (H, W, 3) uint8 at one test scale): ``depth`` sets of a pinned uint8 staging
    buffer, its device copy and pinned result buffers, one copy stream, a few staging threads.

    Per batch: the frames are copied into the pinned buffer by the staging threads (numpy releases
    the GIL), go to the device as ONE asynchronous uint8 copy on the copy stream, and everything
    else -- the batched device pre-process, network + decode (``run_batch``), the device tail
    (``cn_ctdet_post_process_f32``: inverse affine + class grouping) and the copies of the rows /
    class bounds / f32s range digest into pinned memory -- is enqueued on the launch stream
    without a single host synchronisation.  The host waits for batch i - depth + 1 only when it
    collects it, i.e. while later batches are on the device."""

    def __init__(self, det, B, H, W, scale, depth):
        import concurrent.futures
        opt, dev = det.opt, det.opt.device
        self.det, self.B, self.H, self.W, self.scale, self.depth = det, B, H, W, scale, depth
        self.g = det.input_geometry(H, W, scale)
        g = self.g
        self.resize = (g.scaled_h, g.scaled_w) != (g.src_h, g.src_w)
        self.meta = det._meta(g)
        to_input = get_affine_transform(g.center, g.extent, 0, [g.inp_w, g.inp_h])
        self.dst_to_src = (ctypes.c_double * 6)(*invert_affine(to_input).reshape(-1))
        self.mean = (ctypes.c_float * 3)(*[float(v) for v in det.mean.reshape(-1)])
        self.std = (ctypes.c_float * 3)(*[float(v) for v in det.std.reshape(-1)])
        self.pinned_in = [torch.empty((B, H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.np_in = [t.numpy() for t in self.pinned_in]
        self.dev_in = [torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(depth)]
        self.scaled = torch.empty((B, g.scaled_h, g.scaled_w, 3), dtype=torch.uint8, device=dev) \
            if self.resize else None
        self.batch = torch.empty((B, 3, g.inp_h, g.inp_w), device=dev, dtype=torch.float32)
        self.copy_stream = torch.cuda.Stream()
        self.ev_h2d = [torch.cuda.Event() for _ in range(depth)]
        self.ev_pre = [torch.cuda.Event() for _ in range(depth)]
        self.ev_done = [torch.cuda.Event() for _ in range(depth)]
        self.used = [False] * depth
        self.digest_host = [torch.zeros(2, dtype=torch.int32).pin_memory() for _ in range(depth)]
        self.has_digest = [False] * depth
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=min(4, B))
        self.tail = det._device_tail_alloc(self) if det._device_tail_alloc is not None else None
        self.dets_host = None if self.tail is not None else [None] * depth

    def _stage(self, slot, frames):
        dst = self.np_in[slot]
        n = len(frames)
        step = -(-n // self.pool._max_workers)

        def copy(lo):
            for i in range(lo, min(lo + step, n)):
                np.copyto(dst[i], frames[i])
        list(self.pool.map(copy, range(0, n, step)))

    def submit(self, i, frames):
        det, lib, g, B = self.det, native.lib(), self.g, self.B
        slot = i % self.depth
        if self.used[slot]:
            self.ev_h2d[slot].synchronize()      # the pinned buffer's previous upload has left it
        self._stage(slot, frames)
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(self.copy_stream):
            if self.used[slot]:
                self.copy_stream.wait_event(self.ev_pre[slot])   # the device copy's previous reader is done
            self.dev_in[slot].copy_(self.pinned_in[slot], non_blocking=True)
            self.ev_h2d[slot].record(self.copy_stream)
        self.used[slot] = True
        cur.wait_event(self.ev_h2d[slot])
        stream = native.stream_ptr()
        src = self.dev_in[slot]
        if self.resize:
            for j in range(B):
                native.check(lib.cn_resize_bilinear_u8(native.ptr(src[j]), g.src_h, g.src_w, g.src_w * 3,
                                                       g.scaled_h, g.scaled_w, native.ptr(self.scaled[j]), stream),
                             "cn_resize_bilinear_u8")
            src = self.scaled
        native.check(lib.cn_warp_normalize_u8_f32_batch(
            native.ptr(src), B, g.scaled_h * g.scaled_w * 3, g.scaled_h, g.scaled_w, g.scaled_w * 3,
            self.dst_to_src, g.inp_h, g.inp_w, self.mean, self.std, 0, native.ptr(self.batch), stream),
            "cn_warp_normalize_u8_f32_batch")
        self.ev_pre[slot].record()
        dets = det.run_batch(self.batch)
        plan = det.model.plan_for(B, g.inp_h, g.inp_w, self.batch.device)
        rs = getattr(plan.b, "range_sum", None) if plan.b.range is not None else None
        self.has_digest[slot] = rs is not None
        if rs is not None:
            self.digest_host[slot].copy_(rs, non_blocking=True)
        if self.tail is not None:
            det._device_tail_run(self, slot, dets)
        else:
            self.dets_host[slot] = torch.empty(dets.shape, dtype=dets.dtype).pin_memory() \
                if self.dets_host[slot] is None else self.dets_host[slot]
            self.dets_host[slot].copy_(dets, non_blocking=True)
        self.ev_done[slot].record()

    def collect(self, i, frames):
        """Results of batch i (waits for it; later batches keep the device busy)."""
        from ..engine import F16_MAX_BITS
        det = self.det
        slot = i % self.depth
        self.ev_done[slot].synchronize()
        det.__dict__["_unchecked"] = 0       # (the batch's range digest is looked at right here)
        if self.has_digest[slot] and (int(self.digest_host[slot][0]) & 0xffffffff) > F16_MAX_BITS:
            # an f32s value was clamped somewhere up to this batch: results invalid.  Drain the
            # device, let the module re-calibrate, and run this batch again synchronously.
            torch.cuda.synchronize()
            det.range_ok(None)
            return det._run_frames_sync(frames, self.scale)
        metas = [self.meta] * len(frames)
        if self.tail is not None:
            return det._device_tail_results(self, slot, len(frames))
        return det.results_batch(self.dets_host[slot].numpy()[:len(frames)], metas, self.scale)


class BaseDetector(object):
    def __init__(self, opt):
        if opt.gpus[0] < 0:
            raise native.NativeError("--gpus -1 (CPU) is not supported: centernet_amd is the "
                                     "MI355X path and has no CPU fallback")
        opt.device = torch.device('cuda')
        print('Creating model...')
        net = create_model(opt.arch, opt.heads, opt.head_conv)
        if opt.load_model:
            net = load_model(net, opt.load_model)
        self.model = net.to(opt.device).eval()
        if getattr(opt, 'fp32_mfma', False):
            self.model.fp32_mfma()
        self.opt = opt
        self.mean = np.asarray(opt.mean, np.float32).reshape(1, 1, 3)
        self.std = np.asarray(opt.std, np.float32).reshape(1, 1, 3)
        self.scales = opt.test_scales
        self.num_classes = opt.num_classes
        self.max_per_image = 100
        self.pause = True

    # ------------------------------------------------------------------ input geometry
    def input_geometry(self, height, width, scale):
        """Network input size, centre and extent for one test scale (base_detector.py:38-50):
        the fixed resolution, or the scaled image size rounded up to a multiple of pad + 1."""
        scaled_h, scaled_w = int(height * scale), int(width * scale)
        if self.opt.fix_res:
            inp_h, inp_w = self.opt.input_h, self.opt.input_w
            center = np.array([scaled_w / 2., scaled_h / 2.], dtype=np.float32)
            extent = max(height, width) * 1.0
        else:
            inp_h, inp_w = (scaled_h | self.opt.pad) + 1, (scaled_w | self.opt.pad) + 1
            center = np.array([scaled_w // 2, scaled_h // 2], dtype=np.float32)
            extent = np.array([inp_w, inp_h], dtype=np.float32)
        return InputGeometry(height, width, scaled_h, scaled_w, inp_h, inp_w, center, extent)

    def _meta(self, g):
        return {'c': g.center, 's': g.extent,
                'out_height': g.inp_h // self.opt.down_ratio,
                'out_width': g.inp_w // self.opt.down_ratio}

    # ------------------------------------------------------------------ pre-process
    def pre_process(self, image, scale, meta=None):
        """Host form (base_detector.py:37-65): resize, affine warp, normalise, CHW, flip concat.
        Usable from DataLoader workers; returns a CPU tensor and the meta dict."""
        g = self.input_geometry(image.shape[0], image.shape[1], scale)
        to_input = get_affine_transform(g.center, g.extent, 0, [g.inp_w, g.inp_h])
        warped = warp_affine(resize_bilinear(image, (g.scaled_w, g.scaled_h)), to_input,
                             (g.inp_w, g.inp_h))
        batch = normalize_chw(warped, self.mean, self.std)[None]
        if self.opt.flip_test:
            batch = np.concatenate((batch, batch[:, :, :, ::-1]), axis=0)
        return torch.from_numpy(np.ascontiguousarray(batch)), self._meta(g)

    def pre_process_device(self, image, scale, meta=None, out=None):
        """The same steps on the device: the uint8 frame (a numpy array, or a uint8 HIP tensor
        that is already uploaded) goes through ``cn_resize_bilinear_u8`` (scale != 1) and
        ``cn_warp_normalize_u8_f32``; the fp32 (1|2,3,H,W) batch is produced in HBM -- into
        ``out`` when given.  Bit-identical to ``pre_process``."""
        lib = native.lib()
        dev = self.opt.device
        if torch.is_tensor(image):
            if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3 or \
                    not image.is_cuda or not image.is_contiguous():
                raise ValueError("pre_process_device needs a contiguous (H, W, 3) uint8 HIP tensor")
            frame = image
        else:
            if image.dtype != np.uint8 or image.ndim != 3 or image.shape[2] != 3:
                raise ValueError("pre_process_device needs an (H, W, 3) uint8 BGR image")
            frame = torch.from_numpy(np.ascontiguousarray(image)).to(dev)
        g = self.input_geometry(int(frame.shape[0]), int(frame.shape[1]), scale)
        stream = native.stream_ptr()
        if (g.scaled_h, g.scaled_w) != (g.src_h, g.src_w):
            scaled = torch.empty((g.scaled_h, g.scaled_w, 3), device=dev, dtype=torch.uint8)
            native.check(lib.cn_resize_bilinear_u8(native.ptr(frame), g.src_h, g.src_w, g.src_w * 3,
                                                   g.scaled_h, g.scaled_w, native.ptr(scaled), stream),
                         "cn_resize_bilinear_u8")
            frame = scaled
        shape = (2 if self.opt.flip_test else 1, 3, g.inp_h, g.inp_w)
        if out is None:
            out = torch.empty(shape, device=dev, dtype=torch.float32)
        elif tuple(out.shape) != shape or not out.is_contiguous():
            raise ValueError("pre_process_device: `out` must be a contiguous %s tensor" % (shape,))
        to_input = get_affine_transform(g.center, g.extent, 0, [g.inp_w, g.inp_h])
        dst_to_src = (ctypes.c_double * 6)(*invert_affine(to_input).reshape(-1))
        mean = (ctypes.c_float * 3)(*[float(v) for v in self.mean.reshape(-1)])
        std = (ctypes.c_float * 3)(*[float(v) for v in self.std.reshape(-1)])
        native.check(lib.cn_warp_normalize_u8_f32(native.ptr(frame), g.scaled_h, g.scaled_w,
                                                  g.scaled_w * 3, dst_to_src, g.inp_h, g.inp_w,
                                                  mean, std, int(self.opt.flip_test),
                                                  native.ptr(out), stream),
                     "cn_warp_normalize_u8_f32")
        return out, self._meta(g)

    # ------------------------------------------------------------------ task hooks
    def process(self, images, return_time=False):
        raise NotImplementedError

    def post_process(self, dets, meta, scale=1):
        raise NotImplementedError

    def merge_outputs(self, detections):
        raise NotImplementedError

    def _no_debugger(self):
        # the reference's Debugger (utils/debugger.py: cv2 / matplotlib windows) is outside the hot
        # path and not built: --debug >= 1 runs the detector as --debug 0 does and says so once
        if not self.__dict__.get("_debug_warned"):
            self.__dict__["_debug_warned"] = True
            import warnings
            warnings.warn("--debug %d: visual debugging is not part of centernet_amd; results are "
                          "computed and returned as with --debug 0" % self.opt.debug)

    def debug(self, debugger, images, dets, output, scale=1):
        self._no_debugger()

    def show_results(self, debugger, image, results):
        self._no_debugger()

    # ------------------------------------------------------------------ run
    @staticmethod
    def _read_bgr(path):
        from PIL import Image  # cv2.imread replacement: BGR uint8
        rgb = np.asarray(Image.open(path).convert('RGB'))
        return np.ascontiguousarray(rgb[:, :, ::-1])

    def _inputs_for_scale(self, image, prefetched, scale, meta):
        """(images on the device, meta) of one test scale."""
        if prefetched is not None:   # test.py's PrefetchDataset dict (base_detector.py:104-110)
            meta = {k: v.numpy()[0] for k, v in prefetched['meta'][scale].items()}
            return prefetched['images'][scale][0].to(self.opt.device), meta
        on_device = getattr(self.opt.device, 'type', str(self.opt.device)) == 'cuda'
        if on_device and image.dtype == np.uint8 and not getattr(self.opt, 'host_pre_process', False):
            return self.pre_process_device(image, scale, meta)
        images, meta = self.pre_process(image, scale, meta)
        return images.to(self.opt.device), meta

    def results_batch(self, dets, metas, scale):
        """Host tail of ``run_frames`` (task specific): host array of raw detections + the
        frames' metas -> what ``run(frame)['results']`` returns, per image."""
        raise NotImplementedError

    # device tail of the frame pipeline: task classes that have one set the three hooks
    _device_tail_alloc = None

    def _run_frames_sync(self, frames, scale):
        """One batch, synchronously, frame by frame through ``pre_process_device`` (the comparison
        path of the pipeline, and its re-run path after an f32s re-calibration)."""
        uploaded = torch.from_numpy(np.ascontiguousarray(np.stack(frames))).to(self.opt.device)
        g = self.input_geometry(uploaded.shape[1], uploaded.shape[2], scale)
        batch = torch.empty((len(frames), 3, g.inp_h, g.inp_w), device=self.opt.device,
                            dtype=torch.float32)
        metas = [self.pre_process_device(frame, scale, out=batch[i:i + 1])[1]
                 for i, frame in enumerate(uploaded)]
        dets = self.run_batch(batch).detach().cpu().numpy()
        if not self.range_ok(batch):    # a clamped f32s value: re-calibrated on this batch, run again
            dets = self.run_batch(batch).detach().cpu().numpy()
            if not self.range_ok(batch):
                raise native.NativeError("f32s forward clamps values after re-calibration")
        return self.results_batch(dets, metas, scale)

    def _pipe_for(self, frames, depth):
        if len(self.scales) != 1 or self.opt.flip_test:
            raise ValueError("run_frames is single-scale, no flip")
        if getattr(self.opt, "nms", False):
            # merge_outputs applies soft-NMS under --nms (detectors/ctdet.py:63-64); the batched tail
            # does not: refuse rather than return something else than run(frame)['results']
            raise ValueError("run_frames does not apply --nms (soft-NMS): use run(frame)")
        shapes = {tuple(f.shape) for f in frames}
        if len(shapes) != 1:
            raise ValueError("run_frames needs frames of one size")
        (H, W, C), = shapes
        if C != 3 or any(f.dtype != np.uint8 for f in frames):
            raise ValueError("run_frames needs (H, W, 3) uint8 BGR frames")
        key = (len(frames), H, W, self.scales[0], depth)
        pipes = self.__dict__.setdefault("_pipes", {})
        if key not in pipes:
            if len(pipes) >= 4:
                pipes.pop(next(iter(pipes))).pool.shutdown(wait=False)
            pipes[key] = _FramePipe(self, len(frames), H, W, self.scales[0], depth)
        return pipes[key]

    def run_frames(self, frames):
        """A list of (H, W, 3) uint8 BGR frames of one size -> list of per-image results, what
        ``run(frame)['results']`` returns for each (single scale, no flip).  The reference's
        test loop is batch_size = 1 (test.py:60-62); here the frames are uploaded as ONE uint8
        copy, pre-processed on the device in one launch straight into one batch tensor, the whole
        batch goes through the network + decode once (``run_batch``) and, for ctdet, through the
        device tail (inverse affine + class grouping); the host slices the result."""
        pipe = self._pipe_for(frames, 1)
        pipe.submit(0, frames)
        return pipe.collect(0, frames)

    def run_frames_stream(self, batches, depth=3):
        """``run_frames`` over an iterable of batches (lists of frames, all batches of one size and
        frame geometry), pipelined: while batch i is on the device the host stages batch i + 1
        (pinned uint8 copy by a few threads, asynchronous upload on a copy stream) and builds the
        result dictionaries of batch i - 1.  Yields the per-image results batch by batch, in order."""
        pipe, pending = None, collections.deque()
        n = 0
        for frames in batches:
            if pipe is None:
                pipe = self._pipe_for(frames, depth)
            elif (len(frames), ) + tuple(frames[0].shape) != (pipe.B, pipe.H, pipe.W, 3):
                raise ValueError("run_frames_stream needs batches of one size and frame geometry")
            if len(pending) == depth:
                j, fr = pending.popleft()
                yield pipe.collect(j, fr)
            pipe.submit(n, frames)
            pending.append((n, frames))
            n += 1
        while pending:
            j, fr = pending.popleft()
            yield pipe.collect(j, fr)

    UNCHECKED_LIMIT = 4096     # run_batch forwards without a look at the range words before a warning

    def _note_unchecked_forward(self):
        """``run_batch`` does not look at the f32s range words (it never synchronises): the caller
        owes a ``range_ok()`` where it consumes results.  A caller that never pays gets told, once."""
        n = self.__dict__.get("_unchecked", 0) + 1
        self.__dict__["_unchecked"] = n
        if n == self.UNCHECKED_LIMIT and self.model.uses_f32s():
            import warnings
            warnings.warn("%d run_batch() forwards without a range_ok() look: a clamped f32s value "
                          "would go unnoticed (call detector.range_ok() where results are consumed)" % n)

    def range_ok(self, images=None):
        """Synchronising look at the f32s range words of every forward since the last look
        (``PlannedModule.range_ok``): False = a value was clamped, those results are invalid and
        the network has been re-calibrated (on ``images`` when given) -- run the batch again."""
        self.__dict__["_unchecked"] = 0
        return self.model.range_ok(images)

    def run(self, image_or_path_or_tensor, meta=None):
        """One image (array, path, or the prefetch dict) through every test scale
        (base_detector.py:82-143, debug == 0 path); returns the results and the time buckets."""
        clock = _PhaseClock()
        prefetched = None
        if isinstance(image_or_path_or_tensor, np.ndarray):
            image = image_or_path_or_tensor
        elif isinstance(image_or_path_or_tensor, str):
            image = self._read_bgr(image_or_path_or_tensor)
        else:
            prefetched = image_or_path_or_tensor
            image = prefetched['image'][0].numpy()
        clock.lap("load", sync=False)

        per_scale = []
        for scale in self.scales:
            images, meta = self._inputs_for_scale(image, prefetched, scale, meta)
            clock.lap("pre")
            output, dets, forward_done = self.process(images, return_time=True)
            clock.lap("net", until=forward_done)   # process() took this stamp after its own sync
            clock.lap("dec")
            if self.opt.debug >= 2:
                self.debug(None, images, dets, output, scale)
            per_scale.append(self.post_process(dets, meta, scale))
            clock.lap("post")
        results = self.merge_outputs(per_scale)
        clock.lap("merge")
        if self.opt.debug >= 1:
            self.show_results(None, image, results)
        out = {'results': results}
        out.update(clock.finish())
        return out
