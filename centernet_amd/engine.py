"""Network plan: turns a CenterNet module tree into a fixed sequence of HIP launches.

The ``nn.Module`` classes under ``centernet_amd/networks`` only *hold parameters*
(with the reference's state-dict names, so zoo checkpoints load).  Their forward
pass is not a chain of torch ops: each network describes itself once to a
``PlanBuilder`` (conv+BN+ReLU(+residual) -> one implicit-GEMM launch, DCN+BN+ReLU ->
one fused deformable launch, ...), weights are re-packed for the MFMA kernels, BN
(eval) is folded to a per-channel fp32 scale/shift epilogue, activations live in
NHWC, and the resulting launch list is replayed per batch on the current HIP stream
(optionally captured in a HIP graph).

Reference call sites being replaced: ``self.model(images)[-1]``
(src/lib/detectors/ctdet.py:30) and everything below it
(src/lib/models/networks/*.py ``forward``).
"""
import ctypes
import math
import os
import warnings

import torch

from . import native
from .native import (ConvDesc, F32sCtl, LAYOUT_NCHW, LAYOUT_NHWC, DTYPE_F16, DTYPE_F32, DTYPE_F32S,
                     CONV_X_PLAIN, CONV_Y_PLAIN, CONV_R_PLAIN, CONV_STEM_F32S,
                     CONV_STEM_MAXPOOL, CONV_STEM_Y_F32S)

# ---- f32s range policy (csrc/cn_common.h "Range") ------------------------------------------
# A tensor of exponent e is stored as real * 2^-e.  e is chosen on a plain-fp32 calibration pass
# so that the tensor's largest magnitude lands in [2^(TOP_LOG2-1), 2^TOP_LOG2): 2^6 of head-room
# below the fp16 limit, and the (high, low) pair keeps its 22 bits down to values 2^-12 of that
# maximum (absolute error floor 2^-25 stored = 2^-34 of the maximum).
TOP_LOG2 = 10
F16_MAX = 65504.0
LOW_WATER = 2.0 ** -3      # a per-forward maximum below this (stored units) asks for re-calibration
LOW_STEP = 6               # ... which lowers an exponent by at most this many binades at a time
LOW_EVERY = 64             # ... and happens at most once per this many forwards
F16_MAX_BITS = 0x477FE000  # float32 bit patterns of the two bounds (non-negative floats order like
LOW_WATER_BITS = 0x3E000000  # unsigned integers: the range words are compared as bits)
W_TOP_LOG2 = 14            # weight rows are pre-scaled to max |w| in [2^13, 2^14)


def exponent_for(absmax):
    """Exponent e with absmax * 2^-e in [2^(TOP_LOG2-1), 2^TOP_LOG2) (0 for an all-zero tensor)."""
    if math.isinf(absmax) or math.isnan(absmax):
        raise native.NativeError("non-finite activations in the f32s calibration pass")
    if not absmax > 0.0:
        return 0
    return math.frexp(absmax)[1] - TOP_LOG2


def _out_size(n, k, s, p, d=1):
    return (n + 2 * p - (d * (k - 1) + 1)) // s + 1


class Act:
    """An activation: NHWC tensor ``t`` of shape (B,H,W,pitch) using channels
    [c_off, c_off + C).  For NCHW tensors (network input, head outputs) ``nchw`` is set.
    ``fmt``: "f32" plain floats, "f16" halves, or "f32s" -- fp32 values stored as fp16
    (high, low) pairs in 128-byte groups of 32 channels (same bytes and pitch as fp32; the
    tensor is typed float32 but opaque to torch), see csrc/cn_common.h."""
    __slots__ = ("t", "B", "H", "W", "C", "pitch", "c_off", "nchw", "fmt", "exp", "lid")

    def __init__(self, t, B, H, W, C, pitch=None, c_off=0, nchw=False, fmt="f32", exp=0, lid=None):
        self.t, self.B, self.H, self.W, self.C = t, B, H, W, C
        self.pitch = C if pitch is None else pitch
        self.c_off = c_off
        self.nchw = nchw
        self.fmt = fmt
        # ``exp``: the tensor's f32s exponent -- an f32s tensor holds real * 2^-exp; a plain one
        # holds real values and ``exp`` is what a consumer applies when it splits them.
        # ``lid``: logical id of the tensor in the network description (the same in every plan
        # of a module, whatever launches the plan fuses): the key of the calibrated exponents.
        self.exp = exp
        self.lid = lid

    @property
    def applied_exp(self):
        return self.exp if self.fmt == "f32s" else 0

    def ptr(self):
        return ctypes.c_void_p(self.t.data_ptr() + self.t.element_size() * self.c_off)

    def to_float(self):
        """The activation's values as a plain (B, H, W, C) fp32 tensor (host-side decoding with
        torch ops: for tests and debugging, not on the hot path)."""
        t = self.t
        if self.nchw:
            return t.permute(0, 2, 3, 1).float()
        if self.fmt == "f32s":
            h = t.view(torch.float16).reshape(self.B, self.H, self.W, self.pitch // 32, 2, 32)
            t = (h[..., 0, :].float() + h[..., 1, :].float()).reshape(self.B, self.H, self.W,
                                                                      self.pitch)
            t = torch.ldexp(t, torch.tensor(self.exp))
        return t[..., self.c_off:self.c_off + self.C].float()


def prescale_rows(w):
    """f32s weights: a per-output-channel power of two brings max|w| of every row into
    [2^13, 2^14): the fp16 (high, low) pair of a weight then keeps its 22 bits down to 2^-16 of
    the row's largest weight (at [0.5, 1) the low parts went subnormal below 2^-3 of it).  Exact,
    and undone by the returned factors in the epilogue scale.
    Returns (scaled weight, factor per row): w = scaled * factor."""
    w = w.detach().float()
    m = w.abs().flatten(1).amax(dim=1)
    e = torch.frexp(m.clamp_min(1e-30))[1] - W_TOP_LOG2     # m = mantissa * 2^(e + 14)
    e = torch.where(m > 0, e, torch.zeros_like(e))          # an all-zero row keeps factor 1
    shape = (-1,) + (1,) * (w.dim() - 1)
    return torch.ldexp(w, (-e).view(shape).expand_as(w)), torch.ldexp(torch.ones_like(m), e)


def _pow2(e):
    return math.ldexp(1.0, int(e))


def fold_bn(conv_bias, bn, cout, device):
    """BatchNorm2d(eval) [+ conv bias] -> per-channel (scale, shift), fp32.

    y = (acc + bias - mean) * gamma / sqrt(var + eps) + beta
      =  acc * scale + shift
    """
    if bn is None:
        if conv_bias is None:
            return None, None
        return None, conv_bias.detach().to(device=device, dtype=torch.float32).contiguous()
    g = bn.weight.detach().float() if bn.weight is not None else torch.ones(cout)
    b = bn.bias.detach().float() if bn.bias is not None else torch.zeros(cout)
    mean = bn.running_mean.detach().float()
    var = bn.running_var.detach().float()
    scale = g / torch.sqrt(var + bn.eps)
    shift = b - mean * scale
    if conv_bias is not None:
        shift = shift + conv_bias.detach().float() * scale
    return (scale.to(device).contiguous(), shift.to(device).contiguous())


class PlanBuilder:
    """Records launches for one (B, H, W) input shape."""

    RANGE_LAUNCHES = 256   # launches per chunk of range words (cn_f32s_ctl.range); a plan takes as many chunks as it needs
    RANGE_WORDS = 2 * 64 * 16   # CN_RANGE_WORDS: 2 sides x 64 slots x one word per 64-byte line

    def __init__(self, device, B, H, W, dtype=torch.float32, wcache=None, split=None, exps=None,
                 calibrating=False, track=None):
        assert dtype in (torch.float32, torch.float16)
        self.device = device
        # fp32 networks compute in f32s (three fp16 MFMAs per product, fp32-level accuracy, 5.3x
        # the matrix rate of v_mfma_f32_32x32x2_f32) unless split=False / CN_F32S=0 asks for the
        # plain fp32 matrix instruction
        if split is None:
            split = os.environ.get("CN_F32S", "1") != "0"
        self.split = bool(split) and dtype == torch.float32 and not calibrating
        # calibrating: a plain-fp32 plan that materialises every logical tensor (no fused heads)
        # so that PlannedModule.calibrate can measure max |x| of each
        self.calibrating = calibrating
        # logical id -> f32s exponent (missing = 0: values are split as they are, the round-2
        # behaviour, accurate for O(1) tensors only)
        self.exps = dict(exps) if exps else {}
        # packed weights shared by every plan of one module (keyed by source storage, packing
        # and dtype): a new input shape re-uses them instead of re-packing the whole network
        self.wcache = wcache if wcache is not None else {}
        self.dtype = dtype           # element type of NHWC activations and packed weights
        self.cdtype = DTYPE_F16 if dtype == torch.float16 else DTYPE_F32
        self.B, self.H, self.W = B, H, W
        self.lib = native.lib()
        self.ops = []          # list of zero-arg callables
        self.meta = []         # per op: dict(kind, flops, bytes) -- algorithmic work
        self.keep = []         # tensors that must stay alive (weights, descriptors)
        self.flops = 0         # algorithmic conv/DCN FLOPs per forward (2*MACs)
        self.input = None
        self.trace = []        # (kind, Act) of every op output, in launch order (debugging)
        self.ws = None         # split-K scratch shared by all launches (stream-ordered)
        self.fuse_heads = os.environ.get("CN_FUSE_HEADS", "1") != "0" and not calibrating
        self.ws_bytes = 0
        self._nlid = 0
        self.groups = []       # lists of logical ids that must share one exponent (concat)
        # range words of the launches that split values: [cur | hi | lo] x RANGE_WORDS, int32
        # views of float bit patterns (non-negative floats order like integers)
        self.range = None
        self.range_chunks = [] # (cur, stat) tensors per RANGE_LAUNCHES launches
        self._plain_cache = {} # (storage, slice) -> plain copy of an f32s activation (PlanBuilder.plain)
        self.range_slots = []  # slot -> logical id of the launch, for messages
        self._pack_events = [] # completion of weight-pack kernels this plan depends on
        # range words at every split site (csrc/cn_common.h "Range"); ``track=False`` (or CN_RANGE=0)
        # builds the plan WITHOUT them: values beyond the fp16 range are then clamped silently --
        # measurement only (bench.py range_tracking_off_leg)
        if track is None:
            track = os.environ.get("CN_RANGE", "1") != "0"
        self.track = self.split and bool(track)

    # ---- helpers -------------------------------------------------------------
    def _lid(self, lid=None, tag=""):
        """Logical id of the tensor an op produces: a counter over the public op calls of the
        network description (identical in every plan of a module) unless the caller names it."""
        if lid is not None:
            return lid
        self._nlid += 1
        return "t%d%s" % (self._nlid, tag)

    def _exp(self, lid):
        return int(self.exps.get(lid, 0)) if self.split else 0

    def _ctl(self, lid, x_mul=1.0, res_mul=1.0):
        """A cn_f32s_ctl with fresh range words for one launch (NULL words when tracking is off)."""
        c = F32sCtl()
        c.x_mul, c.res_mul = float(x_mul), float(res_mul)
        c.range = None
        if self.track:
            slot = len(self.range_slots)
            ci, si = divmod(slot, self.RANGE_LAUNCHES)
            if ci == len(self.range_chunks):
                # range words grow in chunks of RANGE_LAUNCHES launches -- cur: per launch CN_RANGE_WORDS;
                # stat: [hi | lo] x (launch, side) -- folded chunk by chunk into ONE digest
                cur = torch.zeros((self.RANGE_LAUNCHES, self.RANGE_WORDS), device=self.device, dtype=torch.int32)
                stat = torch.zeros((2, self.RANGE_LAUNCHES, 2), device=self.device, dtype=torch.int32)
                stat[1].fill_(0x7f800000)
                self.range_chunks.append((cur, stat))
                if self.range is None:
                    self.range, self.range_stat = cur, stat
                    # sticky digest of the tables (cn_range_fold_digest) + its pinned host mirror: what
                    # the per-forward check reads (8 bytes, no table walk)
                    self.range_sum = torch.tensor([0, 0x7f800000], device=self.device, dtype=torch.int32)
                    self.range_sum_host = torch.tensor([0, 0x7f800000], dtype=torch.int32).pin_memory()
                    self.range_sum_event = torch.cuda.Event()
            self.range_slots.append(lid)
            c.range = self.range_chunks[ci][0].data_ptr() + 4 * self.RANGE_WORDS * si
        self.keep.append(c)
        return c

    def _new(self, B, H, W, C, pitch=None, fmt=None, lid=None):
        if fmt is None:
            fmt = "f16" if self.dtype == torch.float16 else "f32"
        pitch = C if pitch is None else pitch
        if fmt == "f32s":
            pitch = (pitch + 31) // 32 * 32
            # zero-filled once: the pad channels of the last group are read as K by the next layer
            t = torch.zeros((B, H, W, pitch), device=self.device, dtype=torch.float32)
        else:
            t = torch.empty((B, H, W, pitch), device=self.device, dtype=self.dtype)
        return Act(t, B, H, W, C, pitch, fmt=fmt, exp=self._exp(lid), lid=lid)

    def plain(self, x):
        """``x`` as a plain fp32 activation (a converter launch when it is f32s)."""
        if x is None or x.fmt != "f32s":
            return x
        assert x.c_off % 32 == 0      # a slice starts on a 32-channel group
        # (activations are written once: the plain copy of a tensor serves every later reader)
        key = (id(x.t), x.c_off, x.C)
        hit = self._plain_cache.get(key)
        if hit is not None and hit[0] is x.t:
            return hit[1]
        out = self._new(x.B, x.H, x.W, x.C, fmt="f32", lid=x.lid)
        out.exp = x.exp
        lib, npix = self.lib, x.B * x.H * x.W
        mul = _pow2(x.exp)

        def run():
            rc = lib.cn_f32s_to_f32_scaled(x.ptr(), out.ptr(), npix, x.C, x.pitch, out.pitch, mul,
                                           native.stream_ptr())
            if rc:
                native.check(rc, "cn_f32s_to_f32")
        self._emit_simple(run, "convert", out, 8 * npix * x.C)
        self._plain_cache[key] = (x.t, out)
        return out

    def packed(self, x):
        """``x`` as an f32s activation (a converter launch when it is plain fp32)."""
        if x is None or x.fmt == "f32s":
            return x
        assert x.fmt == "f32" and not x.nchw
        out = self._new(x.B, x.H, x.W, x.C, fmt="f32s", lid=x.lid)
        out.exp = x.exp
        lib, npix = self.lib, x.B * x.H * x.W
        src = Act(x.t, x.B, x.H, x.W, x.C, x.pitch, x.c_off)
        ctl = self._ctl(x.lid)
        mul = _pow2(-x.exp)

        def run():
            rc = lib.cn_f32_to_f32s_scaled(src.ptr(), out.ptr(), npix, x.C, x.pitch, out.pitch, mul,
                                           ctl.range, native.stream_ptr())
            if rc:
                native.check(rc, "cn_f32_to_f32s")
        self._emit_simple(run, "convert", out, 8 * npix * x.C)
        return out

    def _wkey(self, kind, sources):
        return (kind, self.cdtype, str(self.device)) + tuple(
            (id(t), t._version, t.data_ptr(), tuple(t.shape)) for t in sources)

    def _wput(self, key, hit):
        """Cache a packed weight; packed copies of OLDER versions of the same parameter objects
        (in-place updates bump ``_version``) are dropped -- they can never be hit again."""
        ids = tuple(e[0] for e in key[3:])
        for k in [k for k in self.wcache if k[:3] == key[:3] and tuple(e[0] for e in k[3:]) == ids]:
            del self.wcache[k]
        self.wcache[key] = hit

    def _pack(self, w_oihw, sources=None, f32s=False, prescale=False):
        """Packed copy of a (Cout,Cin,KH,KW) weight.  ``sources``: the parameter tensors the
        weight was assembled from (defaults to the weight itself) -- the cache key.  The pack
        kernel runs on the current stream, which also orders the temporary's release: no
        host synchronisation.  ``f32s``: high/low fp16 form of the row-prescaled weight;
        ``prescale``: row-prescaled but packed as plain fp32 (the stem kernel splits it itself).
        Both return (packed, per-row factor to fold into the epilogue scale)."""
        if sources is None:
            sources = [w_oihw]
        # only module parameters have storage that outlives the plan: temporaries are not cached
        cacheable = all(isinstance(t, torch.nn.Parameter) for t in sources)
        kind = "conv_f32s" if f32s else ("conv_pre" if prescale else "conv")
        key = self._wkey(kind, sources) if cacheable else None
        hit = self.wcache.get(key) if cacheable else None
        if hit is None:
            w = w_oihw.detach().to(device=self.device, dtype=torch.float32).contiguous()
            factor = None
            if f32s or prescale:
                w, factor = prescale_rows(w)
                w = w.contiguous()
            cd = DTYPE_F32S if f32s else self.cdtype
            co, ci, kh, kw = w.shape
            n = self.lib.cn_packed_conv_weight_elems(co, ci, kh, kw, cd)
            wp = torch.empty(n, device=self.device, dtype=torch.float32 if f32s else self.dtype)
            native.check(self.lib.cn_pack_conv_weight(native.ptr(w), native.ptr(wp), co, ci, kh,
                                                      kw, cd, native.stream_ptr()),
                         "cn_pack_conv_weight")
            # a plan built on this stream may first run on another one: the packed weight must
            # be complete before that (the pack kernel is tiny; once per weight and module)
            ev = torch.cuda.Event()
            ev.record()
            hit = (wp, factor, ev)
            if cacheable:
                self._wput(key, hit)
        self.keep += [hit[0], hit[1]]
        self._pack_events.append(hit[2])
        return (hit[0], hit[1]) if (f32s or prescale) else hit[0]

    @staticmethod
    def _narrow_form(kh, kw, padding, dilation, out_nchw, ci, co):
        """DLA's 16-channel 3x3 layers (level0 / level1, pose_dla_dcn.py:237-240): plain fp32 tensors
        at full resolution (an f32s tensor would be padded to 32 channels: twice the bytes) and
        their own kernels (cn_conv16.hip)."""
        return ci == 16 and co <= 32 and (kh, kw, padding, dilation) == (3, 3, 1, 1) and not out_nchw

    @classmethod
    def _f32s_conv_form(cls, kh, kw, stride, padding, dilation, out_nchw, ci=0, co=0):
        """Convolution forms that run in f32s: every NHWC-input form; of the 16-channel layers the
        stride-1 one, and only in f32s ARITHMETIC (plain input split while it is staged, plain
        output: conv16s_kernel; CN_C16_F32S=0 keeps it on the fp32 16x16x4 kernel)."""
        if cls._narrow_form(kh, kw, padding, dilation, out_nchw, ci, co):
            # stride 1 (level0: 0.485 -> 0.334 ms at B = 32); the stride-2 form (level1) is
            # register-bound in f32s (0.245 -> 0.476 ms) and stays on the fp32 kernel
            return stride == 1 and os.environ.get("CN_C16_F32S", "1") != "0"
        return True

    def _grow_ws(self, need):
        # split-K scratch shared by every launch (stream-ordered; ops read self.ws at run time)
        if need > self.ws_bytes:
            self.ws = torch.empty(need, device=self.device, dtype=torch.uint8)
            self.ws_bytes = need

    def set_input(self, C=3):
        """Network input: user NCHW image batch, bound at run time."""
        self.input = Act(None, self.B, self.H, self.W, C, nchw=True, exp=self._exp("input"),
                         lid="input")
        return self.input

    # ---- ops -------------------------------------------------------------------
    def conv(self, x, weight, bias=None, bn=None, relu=False, residual=None, stride=1,
             padding=0, dilation=1, out_nchw=False, out=None, wsources=None, out_plain=False,
             pool=None, lid=None):
        """conv2d (+bias) (+BN eval) (+residual) (+ReLU) as one implicit-GEMM launch.
        ``out_plain``: in an f32s plan, write the result as plain fp32 (for consumers that read
        floats, e.g. the offset maps of the deformable kernel)."""
        co, ci, kh, kw = weight.shape
        assert ci == x.C, (ci, x.C)
        lid = self._lid(lid)
        Ho = _out_size(x.H, kh, stride, padding, dilation)
        Wo = _out_size(x.W, kw, stride, padding, dilation)
        use_s = self.split and not x.nchw and self._f32s_conv_form(kh, kw, stride, padding,
                                                                  dilation, out_nchw, ci, co)
        if use_s and self._narrow_form(kh, kw, padding, dilation, out_nchw, ci, co) and residual is None:
            x, out_plain = self.plain(x), True     # plain in, plain out: only the arithmetic is f32s
        # ``pool`` = (kernel, stride, padding) of a MaxPool2d behind the layer (the ResNet stem,
        # resnet_dcn.py:138-141): inside the stem kernel when it takes the shape, else a second launch
        fuse_pool = False
        if pool is not None and tuple(pool) == (3, 2, 1) and self.split and x.nchw and out is None \
                and not out_nchw and residual is None and os.environ.get("CN_FUSE_STEM_POOL", "1") != "0":
            probe = ConvDesc(B=x.B, H=x.H, W=x.W, Cin=ci, Ho=Ho, Wo=Wo, Cout=co, KH=kh, KW=kw,
                             stride=stride, pad_h=padding, pad_w=padding, dil=dilation,
                             in_layout=LAYOUT_NCHW, out_layout=LAYOUT_NHWC, dtype=self.cdtype)
            fuse_pool = bool(self.lib.cn_stem_maxpool_supported(ctypes.byref(probe)))
        scale, shift = fold_bn(bias, bn, co, self.device)
        flags = 0
        # f32s arithmetic inside the stem kernel (image and weights split there) when the library
        # has that form for the shape; otherwise the plain fp32 stem
        stem_s = False
        if self.split and x.nchw and not use_s:
            probe = ConvDesc(B=x.B, H=x.H, W=x.W, Cin=ci, Ho=Ho, Wo=Wo, Cout=co, KH=kh, KW=kw,
                             stride=stride, pad_h=padding, pad_w=padding, dil=dilation,
                             in_layout=LAYOUT_NCHW, out_layout=LAYOUT_NHWC, dtype=self.cdtype,
                             OH=Ho, OW=Wo, oy_mul=1, ox_mul=1,
                             flags=CONV_STEM_F32S | (CONV_STEM_MAXPOOL if fuse_pool else 0))
            stem_s = bool(self.lib.cn_stem_f32s_supported(ctypes.byref(probe)))
        if use_s:
            wp, factor = self._pack(weight, wsources, f32s=True)
            if x.fmt == "f32":
                flags |= CONV_X_PLAIN
            if residual is not None and residual.fmt == "f32":
                flags |= CONV_R_PLAIN
            cd = DTYPE_F32S
        else:
            x, residual = self.plain(x), self.plain(residual)
            if stem_s:
                wp, factor = self._pack(weight, wsources, prescale=True)
                flags |= CONV_STEM_F32S     # the stem kernel splits image and weights itself
            else:
                wp, factor = self._pack(weight, wsources), None
            cd = self.cdtype
            if fuse_pool:
                flags |= CONV_STEM_MAXPOOL
        pool_lid = lid + "/pool" if pool is not None else None
        out_given = out is not None
        if out is None:
            if out_nchw:
                t = torch.empty((x.B, co, Ho, Wo), device=self.device, dtype=torch.float32)
                out = Act(t, x.B, Ho, Wo, co, nchw=True, lid=lid)
            elif fuse_pool:
                # the pooled stem map goes to f32s layers (the first BasicBlock: conv + residual): written
                # as an f32s tensor it lets them run on the persistent 3x3 kernel (CN_STEM_Y_F32S=0: plain)
                y_s = stem_s and co % 32 == 0 and os.environ.get("CN_STEM_Y_F32S", "1") != "0"
                out = self._new(x.B, Ho // 2, Wo // 2, co, fmt="f32s" if y_s else None, lid=pool_lid)
                if y_s:
                    flags |= CONV_STEM_Y_F32S
            else:
                out = self._new(x.B, Ho, Wo, co,
                                fmt="f32s" if (use_s and not out_plain) else None, lid=lid)
        else:
            out.lid, out.exp = lid, self._exp(lid)
        if use_s and out.fmt != "f32s":
            flags |= CONV_Y_PLAIN
        assert use_s or stem_s or out.fmt != "f32s"
        res_pitch = 0
        if residual is not None and residual.pitch != out.pitch:
            sliced = out.pitch != (out.C if out.fmt != "f32s" else (out.C + 31) // 32 * 32) or \
                residual.pitch != (residual.C if residual.fmt != "f32s" else (residual.C + 31) // 32 * 32)
            if sliced:
                # the output or the residual is a channel slice of a wider tensor (a concatenation
                # buffer): the residual keeps its own pixel pitch where the library takes one
                res_pitch = residual.pitch
            else:
                # channel counts that are not a multiple of 32: an f32s tensor is padded to whole
                # groups, a plain one is not -- bring the residual to the output's format
                assert use_s
                if out.fmt == "f32s":
                    residual = self.packed(residual)
                    flags &= ~CONV_R_PLAIN
                else:
                    residual = self.plain(residual)
                    flags |= CONV_R_PLAIN
                assert residual.pitch == out.pitch
        if res_pitch:
            probe = ConvDesc(B=x.B, H=x.H, W=x.W, Cin=ci, Ho=Ho, Wo=Wo, Cout=co, KH=kh, KW=kw,
                             stride=stride, pad_h=padding, pad_w=padding, dil=dilation,
                             in_layout=LAYOUT_NCHW if x.nchw else LAYOUT_NHWC, in_pitch=x.pitch,
                             out_layout=LAYOUT_NCHW if out.nchw else LAYOUT_NHWC, out_pitch=out.pitch,
                             OH=Ho, OW=Wo, oy_mul=1, oy_add=0, ox_mul=1, ox_add=0, relu=int(relu),
                             dtype=cd, flags=flags, res_pitch=res_pitch)
            if not self.lib.cn_conv2d_res_pitch_supported(ctypes.byref(probe)):
                # this layer's kernel wants the residual at the output's pitch: write a private tensor
                # (the caller's concat copies it into the buffer)
                assert out_given and residual.pitch == (residual.C if residual.fmt != "f32s"
                                                        else (residual.C + 31) // 32 * 32), \
                    "a residual slice needs a kernel that takes its pitch"
                return self.conv(x, weight, bias=bias, bn=bn, relu=relu, residual=residual, stride=stride,
                                 padding=padding, dilation=dilation, out_nchw=out_nchw, out=None,
                                 wsources=wsources, out_plain=out_plain, pool=pool, lid=lid)
        # ---- exponents (csrc/cn_common.h "Range"): the matrix loop sees x * 2^-ex and
        # w / factor; the epilogue returns to the output's stored units
        ctl = None
        if use_s or stem_s:
            ex, ey = x.exp, out.applied_exp
            k = factor * _pow2(ex - ey)
            scale = k if scale is None else scale * k
            if shift is not None:
                shift = shift * _pow2(-ey)
            res_mul = _pow2(residual.applied_exp - ey) if residual is not None else 1.0
            x_mul = _pow2(-ex) if (x.fmt == "f32" or x.nchw) else 1.0
            ctl = self._ctl(lid, x_mul, res_mul)
        scale = None if scale is None else scale.contiguous()
        shift = None if shift is None else shift.contiguous()
        self.keep += [scale, shift]
        d = ConvDesc(B=x.B, H=x.H, W=x.W, Cin=ci, Ho=Ho, Wo=Wo, Cout=co, KH=kh, KW=kw,
                     stride=stride, pad_h=padding, pad_w=padding, dil=dilation,
                     in_layout=LAYOUT_NCHW if x.nchw else LAYOUT_NHWC, in_pitch=x.pitch,
                     out_layout=LAYOUT_NCHW if out.nchw else LAYOUT_NHWC, out_pitch=out.pitch,
                     OH=Ho, OW=Wo, oy_mul=1, oy_add=0, ox_mul=1, ox_add=0, relu=int(relu),
                     dtype=cd, flags=flags)
        if ctl is not None:
            d.ctl = ctl
        if res_pitch:
            d.res_pitch = res_pitch
        fl = 2 * x.B * Ho * Wo * co * ci * kh * kw
        by = 4 * (x.B * x.H * x.W * ci + x.B * out.H * out.W * co * (2 if residual is not None else 1)
                  + co * ci * kh * kw)
        self._emit_conv(d, x, wp, scale, shift, residual, out, dict(kind="conv", flops=fl, bytes=by))
        self.flops += fl
        if pool is not None and not fuse_pool:
            return self.maxpool(out, *pool, lid=pool_lid)
        return out

    def _emit_conv(self, d, x, wp, scale, shift, residual, out, meta):
        lib = self.lib
        self.keep.append(d)
        is_input = x is self.input
        sp, hp = native.ptr(scale), native.ptr(shift)
        rp = residual.ptr() if residual is not None else None
        wpp, op = native.ptr(wp), out.ptr()
        dref = ctypes.byref(d)
        self._grow_ws(lib.cn_conv2d_workspace_bytes(dref))

        def run():
            xp = ctypes.c_void_p(self.input.t.data_ptr()) if is_input else x.ptr()
            wsp = ctypes.c_void_p(self.ws.data_ptr()) if self.ws is not None else None
            rc = lib.cn_conv2d(dref, xp, wpp, sp, hp, rp, op, wsp, self.ws_bytes,
                               native.stream_ptr())
            if rc:
                native.check(rc, "cn_conv2d")
        self.ops.append(run)
        self.meta.append(meta)
        self.trace.append((meta["kind"], out))

    def conv_transpose4x4s2(self, x, weight, bn=None, relu=False, out_plain=False):
        """ConvTranspose2d(k=4, s=2, p=1, bias=False) [+BN+ReLU]: the four output-parity
        2x2 convolutions in one launch (reference: resnet_dcn.py:228-235)."""
        ci, co, kh, kw = weight.shape
        assert (kh, kw) == (4, 4) and ci == x.C
        assert self.dtype == torch.float32, "ConvTranspose is built for fp32 only"
        lib = self.lib
        lid = self._lid()
        use_s = self.split
        scale, shift = fold_bn(None, bn, co, self.device)
        cacheable = isinstance(weight, torch.nn.Parameter)
        key = self._wkey("deconv4x4s2_f32s" if use_s else "deconv4x4s2", [weight])
        hit = self.wcache.get(key) if cacheable else None
        if hit is None:
            w = weight.detach().to(device=self.device, dtype=torch.float32)
            factor = None
            if use_s:   # prescale per OUTPUT channel (dim 1 of torch's (Cin, Cout, 4, 4) layout)
                wt, factor = prescale_rows(w.transpose(0, 1).contiguous())
                w = wt.transpose(0, 1)
            w = w.contiguous()
            wp = torch.empty(lib.cn_packed_deconv4x4s2_weight_floats(ci, co), device=self.device,
                             dtype=torch.float32)
            native.check(lib.cn_pack_deconv4x4s2_weight(native.ptr(w), native.ptr(wp), ci, co,
                                                        DTYPE_F32S if use_s else DTYPE_F32,
                                                        native.stream_ptr()),
                         "cn_pack_deconv4x4s2_weight")
            ev = torch.cuda.Event()
            ev.record()
            hit = (wp, factor, ev)
            if cacheable:
                self._wput(key, hit)
        wp, factor, ev = hit
        self._pack_events.append(ev)
        flags = 0
        out = self._new(x.B, 2 * x.H, 2 * x.W, co, fmt="f32s" if (use_s and not out_plain) else None,
                        lid=lid)
        ctl = None
        if use_s:
            ey = out.applied_exp
            k = factor * _pow2(x.exp - ey)
            scale = (k if scale is None else scale * k).contiguous()
            if shift is not None:
                shift = (shift * _pow2(-ey)).contiguous()
            if x.fmt == "f32":
                flags |= CONV_X_PLAIN
            if out_plain:
                flags |= CONV_Y_PLAIN
            ctl = self._ctl(lid, _pow2(-x.exp) if x.fmt == "f32" else 1.0)
        self.keep += [scale, shift, wp, factor]
        sp, hp, wpp = native.ptr(scale), native.ptr(shift), native.ptr(wp)
        cd = DTYPE_F32S if use_s else DTYPE_F32
        cref = ctypes.byref(ctl) if ctl is not None else None

        def run():
            rc = lib.cn_conv_transpose4x4s2(x.ptr(), wpp, sp, hp, out.ptr(), x.B, x.H, x.W, ci, co,
                                            x.pitch, out.pitch, int(relu), cd, flags, cref,
                                            native.stream_ptr())
            if rc:
                native.check(rc, "cn_conv_transpose4x4s2")
        self.ops.append(run)
        fl = 2 * x.B * (2 * x.H) * (2 * x.W) * co * ci * 4
        by = 4 * (x.B * x.H * x.W * (ci + 4 * co) + co * ci * 16)
        self.meta.append(dict(kind="conv", flops=fl, bytes=by))
        self.trace.append(("deconv", out))
        self.flops += fl
        return out

    def maxpool(self, x, k, s, pad, lid=None, out=None, to_s=False):
        """MaxPool2d.  ``to_s`` / ``out``: an f32s input of whole groups stays f32s -- written into
        ``out`` when given (a channel slice of a concatenation buffer) -- so that its consumers (a
        projection, a Root's concatenation: pose_dla_dcn.py:206-221) read (high, low) pairs."""
        assert self.dtype == torch.float32, "max-pool is built for fp32 only"
        lid = self._lid(lid)
        lib = self.lib
        from_s = x.fmt == "f32s" and x.C % 32 == 0     # read (high, low) pairs
        Ho, Wo = _out_size(x.H, k, s, pad), _out_size(x.W, k, s, pad)
        if from_s and (to_s or (out is not None and out.fmt == "f32s")):
            if out is None:
                out = self._new(x.B, Ho, Wo, x.C, fmt="f32s", lid=lid)
            else:
                assert out.fmt == "f32s" and (out.B, out.H, out.W, out.C) == (x.B, Ho, Wo, x.C)
                out.lid, out.exp = lid, self._exp(lid)
            assert x.c_off % 32 == 0 and out.c_off % 32 == 0
            mul = _pow2(x.exp - out.exp)
            ctl = self._ctl(lid)

            def run():
                rc = lib.cn_maxpool_nhwc_f32s(x.ptr(), out.ptr(), x.B, x.H, x.W, x.C, x.pitch, out.pitch,
                                              k, s, pad, mul, ctl.range, native.stream_ptr())
                if rc:
                    native.check(rc, "cn_maxpool_nhwc_f32s")
            self.ops.append(run)
            self.meta.append(dict(kind="maxpool", flops=0, bytes=4 * x.B * x.C * (x.H * x.W + Ho * Wo)))
            self.trace.append(("maxpool", out))
            return out
        if not from_s:
            x = self.plain(x)
        private = self._new(x.B, Ho, Wo, x.C, lid=lid)
        assert x.pitch == x.C and x.c_off == 0
        mul = _pow2(x.exp) if from_s else 1.0

        def run():
            rc = lib.cn_maxpool_nhwc_scaled(x.ptr(), private.ptr(), x.B, x.H, x.W, x.C, k, s, pad,
                                            DTYPE_F32S if from_s else DTYPE_F32, mul,
                                            native.stream_ptr())
            if rc:
                native.check(rc, "cn_maxpool_nhwc")
        self.ops.append(run)
        self.meta.append(dict(kind="maxpool", flops=0,
                              bytes=4 * x.B * x.C * (x.H * x.W + Ho * Wo)))
        self.trace.append(("maxpool", private))
        return private

    def _emit_simple(self, fn, kind, out, nbytes):
        self.ops.append(fn)
        self.meta.append(dict(kind=kind, flops=0, bytes=nbytes))
        self.trace.append((kind, out))

    def concat_buffer(self, B, H, W, widths):
        """A concatenation buffer allocated BEFORE its members exist, and one Act per member to pass
        as ``out=`` to the producing launch (conv / maxpool): members written in place need no copy.
        f32s plans only, members of whole 32-channel groups; otherwise (None, None) -- ``concat``
        then allocates and copies as before.  Pass the buffer to ``concat(acts, into=buf)``."""
        if not self.split or any(w % 32 for w in widths) or os.environ.get("CN_CONCAT_INPLACE", "1") == "0":
            return None, None
        C = sum(widths)
        buf = self._new(B, H, W, C, fmt="f32s")
        slices, off = [], 0
        for w in widths:
            slices.append(Act(buf.t, B, H, W, w, pitch=C, c_off=off, fmt="f32s"))
            off += w
        return buf, slices

    def concat(self, acts, into=None):
        """torch.cat(acts, 1) (Root.forward, pose_dla_dcn.py:159): channel-slice copies into
        one NHWC buffer; members that already live in their slice of ``into`` (``concat_buffer``)
        are not copied."""
        lid = self._lid()
        if into is not None:
            return self._concat_into(acts, into, lid)
        # the inputs of a concatenation share one exponent (PlannedModule.calibrate gives the
        # whole group the exponent of its largest member)
        self.groups.append([lid] + [a.lid for a in acts])
        # f32s tensors of whole 32-channel groups concatenate as they are: a group is 128 bytes
        # whatever the format, so the slice copy moves (high, low) groups unchanged
        keep_s = all(a.fmt == "f32s" and a.C % 32 == 0 and a.pitch == a.C for a in acts) and \
            len({a.exp for a in acts}) == 1
        if not keep_s:
            acts = [self.plain(a) for a in acts]
        a0 = acts[0]
        C = sum(a.C for a in acts)
        out = self._new(a0.B, a0.H, a0.W, C, fmt="f32s" if keep_s else None, lid=lid)
        if keep_s:
            out.exp = a0.exp
        lib = self.lib
        npix = a0.B * a0.H * a0.W
        off = 0
        for a in acts:
            assert (a.B, a.H, a.W) == (a0.B, a0.H, a0.W) and not a.nchw
            dst = Act(out.t, out.B, out.H, out.W, a.C, pitch=C, c_off=off, fmt=out.fmt,
                      exp=out.exp, lid=a.lid)

            def run(a=a, dst=dst):
                rc = lib.cn_copy_channels_f32(a.ptr(), a.pitch, dst.ptr(), C, npix, a.C,
                                              native.stream_ptr())
                if rc:
                    native.check(rc, "cn_copy_channels_f32")
            self._emit_simple(run, "copy", dst, 8 * npix * a.C)
            off += a.C
        return out

    def _concat_into(self, acts, buf, lid):
        self.groups.append([lid] + [a.lid for a in acts])
        C = buf.pitch
        assert buf.fmt == "f32s" and sum(a.C for a in acts) == C
        exps = {a.exp for a in acts}
        buf.lid = lid
        buf.exp = acts[0].exp
        lib, npix, off = self.lib, buf.B * buf.H * buf.W, 0
        for a in acts:
            assert (a.B, a.H, a.W) == (buf.B, buf.H, buf.W) and not a.nchw
            in_place = a.t is buf.t and a.c_off == off and a.pitch == C and a.fmt == "f32s"
            if not in_place:
                src = a if (a.fmt == "f32s" and a.C % 32 == 0 and a.exp == buf.exp) else None
                if src is None:
                    # a member in another format or with another exponent: through plain floats
                    p = self.plain(a)
                    q = Act(p.t, p.B, p.H, p.W, p.C, p.pitch, p.c_off, fmt="f32", exp=buf.exp, lid=a.lid)
                    src = self.packed(q)
                dst = Act(buf.t, buf.B, buf.H, buf.W, a.C, pitch=C, c_off=off, fmt="f32s", exp=buf.exp, lid=a.lid)

                def run(src=src, dst=dst):
                    rc = lib.cn_copy_channels_f32(src.ptr(), src.pitch, dst.ptr(), C, npix, src.C,
                                                  native.stream_ptr())
                    if rc:
                        native.check(rc, "cn_copy_channels_f32")
                self._emit_simple(run, "copy", dst, 8 * npix * a.C)
            off += a.C
        assert len(exps) == 1 or not self.exps, "members of a concatenation share one exponent"
        return buf

    def dw_deconv(self, x, weight, f, add=None):
        """Depthwise ConvTranspose2d(C, C, 2f, stride f, padding f//2, groups=C) + add
        (IDAUp, pose_dla_dcn.py:370-373, 381-386)."""
        lid = self._lid()
        x, add = self.plain(x), self.plain(add)
        C = x.C
        assert tuple(weight.shape) == (C, 1, 2 * f, 2 * f) and x.pitch == C and x.c_off == 0
        wt = weight.detach().to(device=self.device, dtype=torch.float32)
        wt = wt[:, 0].permute(1, 2, 0).reshape(4 * f * f, C).contiguous()
        self.keep.append(wt)
        out = self._new(x.B, x.H * f, x.W * f, C, lid=lid)
        lib = self.lib
        wp = native.ptr(wt)
        if add is not None:
            assert (add.H, add.W, add.C, add.pitch, add.c_off) == (out.H, out.W, C, C, 0)

        def run():
            rc = lib.cn_dw_conv_transpose_f32(x.ptr(), wp, add.ptr() if add is not None else None,
                                              out.ptr(), x.B, x.H, x.W, C, f, native.stream_ptr())
            if rc:
                native.check(rc, "cn_dw_conv_transpose_f32")
        self._emit_simple(run, "dwdeconv", out,
                          4 * x.B * C * (x.H * x.W + (2 if add is not None else 1) * out.H * out.W))
        return out

    def upsample2x_add(self, x, add=None):
        """nn.Upsample(scale_factor=2) (nearest) + skip add (large_hourglass.py:102-109)."""
        lid = self._lid()
        x, add = self.plain(x), self.plain(add)
        assert x.pitch == x.C and x.c_off == 0
        out = self._new(x.B, 2 * x.H, 2 * x.W, x.C, lid=lid)
        lib = self.lib

        fn = lib.cn_upsample2x_add_f16 if self.dtype == torch.float16 else lib.cn_upsample2x_add_f32

        def run():
            rc = fn(x.ptr(), add.ptr() if add is not None else None, out.ptr(), x.B, x.H, x.W, x.C,
                    native.stream_ptr())
            if rc:
                native.check(rc, "cn_upsample2x_add")
        self._emit_simple(run, "upsample", out, 4 * x.B * x.C * x.H * x.W * (1 + 4 + (4 if add is not None else 0)))
        return out

    def dcn(self, x, dcn_mod, bn=None, relu=False, out_plain=False, om=None, mask_sigmoid=True):
        """DCN (DCNv2/dcn_v2.py:44-70) [+ BatchNorm + ReLU]: conv_offset_mask as an
        implicit-GEMM launch writing 27 channels at pitch 32, then the fused deformable
        kernel (gather + MFMA contraction + bias/BN/ReLU epilogue).  ``om``: a ready
        (B, H, W, >= 27) plain Act of [18 offsets | 9 mask values] instead of the offset
        convolution (DCNv2.forward's explicit offset / mask inputs, dcn_v2.py:35-41;
        ``mask_sigmoid=False`` when the mask values are already probabilities)."""
        assert tuple(dcn_mod.kernel_size) == (3, 3) and dcn_mod.stride == 1 and \
            dcn_mod.padding == 1 and dcn_mod.dilation == 1 and dcn_mod.deformable_groups == 1, \
            "only the 3x3/s1/p1/d1/dg1 DCN that CenterNet instantiates is supported"
        assert self.dtype == torch.float32, "the deformable kernel is fp32 only"
        lid = self._lid()
        com = dcn_mod.conv_offset_mask
        # plain fp32 in REAL units (offsets are pixels, the mask a logit): never an f32s tensor
        x = self.plain(x)
        if om is None:
            om = self._new(x.B, x.H, x.W, 27, pitch=32, lid=lid + "/om")
            self.conv(x, com.weight, bias=com.bias, stride=1, padding=1, out=om, lid=lid + "/om")
        assert om.fmt == "f32" and om.pitch >= 27 and (om.B, om.H, om.W) == (x.B, x.H, x.W)
        msig = int(bool(mask_sigmoid))
        co, ci = dcn_mod.weight.shape[0], dcn_mod.weight.shape[1]
        use_s = self.split
        bias = dcn_mod.bias.detach().to(device=self.device, dtype=torch.float32).contiguous()
        scale, shift = fold_bn(None, bn, co, self.device)
        out = self._new(x.B, x.H, x.W, co, fmt="f32s" if (use_s and not out_plain) else None, lid=lid)
        ctl = None
        if use_s:
            wp, factor = self._pack(dcn_mod.weight, f32s=True)
            ey = out.applied_exp
            # (acc' + bias') * scale':  acc' = acc * 2^-ex / factor
            bias = (bias * _pow2(-x.exp) / factor).contiguous()
            k = factor * _pow2(x.exp - ey)
            scale = (k if scale is None else scale * k).contiguous()
            if shift is not None:
                shift = (shift * _pow2(-ey)).contiguous()
            ctl = self._ctl(lid, _pow2(-x.exp))
        else:
            wp = self._pack(dcn_mod.weight)
        self.keep += [bias, scale, shift]
        lib = self.lib
        assert x.pitch == x.C and x.c_off == 0 and x.fmt == "f32"
        bp, sp, hp, wpp = native.ptr(bias), native.ptr(scale), native.ptr(shift), native.ptr(wp)
        cd = DTYPE_F32S if use_s else DTYPE_F32
        flags = CONV_Y_PLAIN if (use_s and out_plain) else 0
        cref = ctypes.byref(ctl) if ctl is not None else None

        self._grow_ws(lib.cn_dcn_v2_forward_nhwc_workspace_bytes(x.B, ci, x.H, x.W, co))

        def run():
            wsp = ctypes.c_void_p(self.ws.data_ptr()) if self.ws is not None else None
            rc = lib.cn_dcn_v2_forward_nhwc(x.ptr(), wpp, bp, om.ptr(), om.pitch, sp, hp,
                                            out.ptr(), out.pitch, x.B, ci, x.H, x.W, co, msig,
                                            int(relu), cd, flags, cref, wsp, self.ws_bytes,
                                            native.stream_ptr())
            if rc:
                native.check(rc, "cn_dcn_v2_forward_nhwc")
        self.ops.append(run)
        fl = 2 * x.B * x.H * x.W * co * ci * 9
        # activations 4*(Cin + 27 + Cout)*HW per image + weights once per launch
        by = 4 * (x.B * x.H * x.W * (ci + 27 + co) + co * ci * 9 + co)
        self.meta.append(dict(kind="dcn", flops=fl, bytes=by, samples=x.B * x.H * x.W * 9 * ci))
        self.trace.append(("dcn", out))
        self.flops += fl
        return out

    def heads(self, x, head_modules):
        """Per head: conv3x3(F->head_conv)+ReLU+conv1x1(head_conv->classes)
        (resnet_dcn.py:155-177, pose_dla_dcn.py:446-468) or a single 1x1 when head_conv=0."""
        seqs = {n: m for n, m in head_modules.items()}
        if all(isinstance(s, torch.nn.Sequential) for s in seqs.values()):
            return self.heads_from_convs(x, {n: (s[0], s[-1]) for n, s in seqs.items()})
        outs = {}
        lid = self._lid()
        for n, s in seqs.items():
            outs[n] = self.conv(x, s.weight, bias=s.bias, stride=1, padding=s.kernel_size[0] // 2,
                                out_nchw=True, lid=lid + "/" + n)
        return outs

    def heads_from_convs(self, x, pairs):
        """``pairs``: name -> (first conv kxk + ReLU, last conv).  The first convolutions of
        all heads read the same feature map, so they run as ONE launch with concatenated
        output channels; each last conv then reads its channel slice and writes the NCHW
        map the decode consumes."""
        lid = self._lid()
        names = list(pairs.keys())
        firsts = [pairs[n][0] for n in names]
        w = torch.cat([c.weight.detach() for c in firsts], 0)
        b = torch.cat([c.bias.detach() for c in firsts], 0)
        k = firsts[0].kernel_size[0]
        if (self.fuse_heads and self.dtype == torch.float32 and k == 3 and
                not x.nchw and x.c_off == 0 and
                self._fusable_hidden([c.weight.shape[0] for c in firsts],
                                     [pairs[n][1].weight.shape[0] for n in names]) and
                all(c.padding[0] == 1 and c.stride[0] == 1 for c in firsts) and
                all(tuple(pairs[n][1].kernel_size) == (1, 1) for n in names)):
            if len(names) <= 8:
                return self._heads_fused(x, names, pairs, w, b, lid)
            # more heads than one launch takes (the exdet task's nine, opts.py:299-306): groups of up
            # to six (head counts the kernel is exercised with: 3 and 6), each its own fused launch
            # over the same feature map; one hidden exponent (lid)
            outs = {}
            for g in range(0, len(names), 6):
                sub = names[g:g + 6]
                outs.update(self._heads_fused(
                    x, sub, pairs, torch.cat([pairs[n][0].weight.detach() for n in sub], 0),
                    torch.cat([pairs[n][0].bias.detach() for n in sub], 0), lid))
            return {n: outs[n] for n in names}
        hcs = [c.weight.shape[0] for c in firsts]
        # an f32s tensor is addressable in whole 32-channel groups only: hidden widths that are
        # not a multiple of 32 keep the hidden layer in plain floats (slices then start anywhere
        # on a 4-channel boundary); widths that are not a multiple of 4 run head by head
        mid_plain = any(h % 32 for h in hcs)
        outs = {}
        if any(h % 4 for h in hcs):
            for n in names:
                first, last = pairs[n]
                mid = self.conv(x, first.weight, bias=first.bias, relu=True, stride=1, padding=k // 2,
                                out_plain=True, lid=lid + "/hid")
                outs[n] = self.conv(mid, last.weight, bias=last.bias, stride=1,
                                    padding=last.kernel_size[0] // 2, out_nchw=True,
                                    lid=lid + "/" + n)
            return outs
        mid = self.conv(x, w, bias=b, relu=True, stride=1, padding=k // 2,
                        wsources=[c.weight for c in firsts], out_plain=mid_plain, lid=lid + "/hid")
        off = 0
        for n in names:
            first, last = pairs[n]
            hc = first.weight.shape[0]
            sl = Act(mid.t, mid.B, mid.H, mid.W, hc, pitch=mid.pitch, c_off=off, fmt=mid.fmt,
                     exp=mid.exp, lid=mid.lid)
            outs[n] = self.conv(sl, last.weight, bias=last.bias, stride=1,
                                padding=last.kernel_size[0] // 2, out_nchw=True, lid=lid + "/" + n)
            off += hc
        return outs


    @staticmethod
    def _fusable_hidden(hidden, couts):
        """Hidden widths the fused head kernel takes: 64 (one slice) or 128 / 192 / 256 (64-channel
        slices one after the other; their 1x1 outputs accumulate in one 96-row register tile)."""
        hc = hidden[0]
        if any(h != hc for h in hidden) or hc % 64 or not 64 <= hc <= 256:
            return False
        if hc > 64 and (max(couts) > 96 or os.environ.get("CN_FUSE_HEADS_WIDE", "1") == "0"):
            return False
        return True

    def _heads_fused(self, x, names, pairs, w1, b1, lid):
        """All heads as ONE launch (cn_heads3x3_1x1): the hidden channels of a head stay in LDS
        between its 3x3 and its 1x1 convolution, 64 at a time."""
        lib = self.lib
        nh = len(names)
        hc = pairs[names[0]][0].weight.shape[0]
        use_s = self.split
        scale1 = None
        eh = self._exp(lid + "/hid")       # exponent of the hidden tile (lives in LDS only)
        b1 = b1.to(device=self.device, dtype=torch.float32)
        ctl = None
        if use_s:
            wp, factor1 = self._pack(w1, [pairs[n][0].weight for n in names], f32s=True)
            scale1 = (factor1 * _pow2(x.exp - eh)).contiguous()
            b1 = b1 * _pow2(-eh)
            ctl = self._ctl(lid + "/hid", _pow2(-x.exp) if x.fmt == "f32" else 1.0)
        else:
            wp = self._pack(w1, [pairs[n][0].weight for n in names])
        b1 = b1.contiguous()
        cd = DTYPE_F32S if use_s else DTYPE_F32
        flags = CONV_X_PLAIN if (use_s and x.fmt == "f32") else 0
        arr = (native.HeadOut * nh)()
        outs = {}
        fl = 2 * x.B * x.H * x.W * w1.shape[0] * x.C * 9
        by = 4 * (x.B * x.H * x.W * x.C + w1.numel())
        for i, n in enumerate(names):
            last = pairs[n][1]
            co = last.weight.shape[0]
            w2 = last.weight.detach().to(device=self.device, dtype=torch.float32)
            w2 = w2.reshape(co, hc).contiguous()
            osc = None
            if use_s:   # the kernel splits the 1x1 weights too: rows pre-scaled like every weight
                w2, factor2 = prescale_rows(w2)
                w2 = w2.contiguous()
                osc = (factor2 * _pow2(eh)).contiguous()
            b2 = None if last.bias is None else \
                last.bias.detach().to(device=self.device, dtype=torch.float32).contiguous()
            t = torch.empty((x.B, co, x.H, x.W), device=self.device, dtype=torch.float32)
            outs[n] = Act(t, x.B, x.H, x.W, co, nchw=True, lid=lid + "/" + n)
            wfrag = None
            if use_s and hc == 64 and co <= 96:
                # the same (prescaled) matrix as MFMA-ready (high, low) fragments: lets the library
                # run the heads on its persistent kernel
                wfrag = torch.empty(lib.cn_packed_head_w2_bytes(co), device=self.device, dtype=torch.uint8)
                native.check(lib.cn_pack_head_w2_f32s(native.ptr(w2), native.ptr(wfrag), co,
                                                      native.stream_ptr()), "cn_pack_head_w2_f32s")
                ev = torch.cuda.Event()
                ev.record()
                self._pack_events.append(ev)
                arr[i].w_frag = wfrag.data_ptr()
            self.keep += [w2, b2, t, osc, wfrag]
            arr[i].w = w2.data_ptr()
            arr[i].bias = b2.data_ptr() if b2 is not None else None
            arr[i].y = t.data_ptr()
            arr[i].cout = co
            arr[i].oscale = osc.data_ptr() if osc is not None else None
            fl += 2 * x.B * x.H * x.W * co * hc
            by += 4 * (x.B * x.H * x.W * co + co * hc)
        self.keep += [b1, arr, scale1]
        wpp, b1p, s1p = native.ptr(wp), native.ptr(b1), native.ptr(scale1)
        ci = x.C
        cref = ctypes.byref(ctl) if ctl is not None else None

        def run():
            rc = lib.cn_heads3x3_1x1(x.ptr(), x.B, x.H, x.W, ci, x.pitch, wpp, s1p, b1p, hc, nh,
                                     arr, cd, flags, cref, native.stream_ptr())
            if rc:
                native.check(rc, "cn_heads3x3_1x1")
        self.ops.append(run)
        self.meta.append(dict(kind="conv", flops=fl, bytes=by))
        self.trace.append(("heads", outs[names[0]]))
        self.flops += fl
        return outs

    def finish(self):
        """Close the launch list: one tiny launch folds this forward's range words into the
        running (largest, smallest) per-launch maxima the host reads at its next look."""
        if self.range is not None and self.range_slots:
            lib = self.lib
            dig = ctypes.c_void_p(self.range_sum.data_ptr())
            left = len(self.range_slots)
            for cur_t, stat_t in self.range_chunks:
                n = min(left, self.RANGE_LAUNCHES)
                left -= n
                cur = ctypes.c_void_p(cur_t.data_ptr())
                hi, lo = (ctypes.c_void_p(stat_t[i].data_ptr()) for i in range(2))

                def run(cur=cur, hi=hi, lo=lo, n=n):
                    rc = lib.cn_range_fold_digest(cur, hi, lo, dig, n, native.stream_ptr())
                    if rc:
                        native.check(rc, "cn_range_fold_digest")
                self.ops.append(run)
                self.meta.append(dict(kind="range", flops=0, bytes=0))
                self.trace.append(("range", None))

    def range_reset(self):
        """Forget what the range words have seen so far (tables and digest)."""
        for cur_t, stat_t in self.range_chunks:
            cur_t.zero_()
            stat_t[0].zero_()
            stat_t[1].fill_(0x7f800000)
        if self.range is not None:
            fresh = torch.tensor([0, 0x7f800000], dtype=torch.int32)
            self.range_sum.copy_(fresh)
            self.range_sum_host.copy_(fresh)


class Plan:
    """A compiled forward pass for one input shape."""

    def __init__(self, builder, outputs):
        builder.finish()
        self.b = builder
        self.outputs = outputs  # name -> Act (NCHW)
        self.graph = None
        self._static_in = None

    @property
    def flops(self):
        return self.b.flops

    def run(self, images, events=None, event_after=None, borrow=False, digest=False):
        """Replay the launch list.  ``events``: optional list that receives one
        torch.cuda.Event per op boundary (len(ops)+1), recorded on the launch stream; with
        ``event_after`` (a set of op indices) events are recorded only at the start and after
        those ops (segment timing with fewer stream markers).

        Returns fresh head tensors, like the reference's nn.Module.  ``borrow=True`` returns the
        plan's own persistent head buffers instead (zero-copy): they are OVERWRITTEN by the next
        run of this plan (same B, H, W) -- only for callers that consume them at once, as the
        detectors do."""
        if not images.is_cuda:
            raise native.NativeError("input batch must be on a HIP device; there is no CPU path")
        native.require_f32(images)
        images = images.contiguous()
        assert tuple(images.shape) == (self.b.B, self.b.input.C, self.b.H, self.b.W), images.shape
        if self.b._pack_events:
            # weights packed on another stream (or just now): complete before the first launch
            st = torch.cuda.current_stream()
            for ev in self.b._pack_events:
                st.wait_event(ev)
            self.b._pack_events = []
        if self.graph is not None:
            self._static_in.copy_(images)
            self.graph.replay()
        else:
            self.b.input.t = images
            if events is None:
                for op in self.b.ops:
                    op()
            else:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                events.append(e)
                for i, op in enumerate(self.b.ops):
                    op()
                    if event_after is None or i in event_after:
                        e = torch.cuda.Event(enable_timing=True)
                        e.record()
                        events.append(e)
        b = self.b
        if b.range is not None and b.range_slots:
            # the digest's 8 bytes travel behind the forward only when the caller will look at them
            # right away (a copy between launches costs ~10 us of stream time: 0.25 % of a B = 32
            # step); otherwise range_quick() fetches them when asked
            if digest:
                b.range_sum_host.copy_(b.range_sum, non_blocking=True)
                b.range_sum_event.record()
            self._digest_fresh = bool(digest)
        if borrow:
            return {k: v.t for k, v in self.outputs.items()}
        return {k: v.t.clone() for k, v in self.outputs.items()}

    # ---- f32s range words ------------------------------------------------------------------
    def range_report(self, reset=True):
        """Synchronising read of the running range words of this plan since the last look:
        per launch that splits values, (logical id, largest, smallest) of its per-forward
        maxima -- output side and input side, in STORED units (value * 2^-exponent; anything
        above 65504 was clamped).  None when the plan is not f32s or tracking is off."""
        b = self.b
        if b.range is None or not b.range_slots:
            return None
        n = len(b.range_slots)
        RL = b.RANGE_LAUNCHES
        host = torch.cat([st[:, :min(RL, n - i * RL)] for i, (_, st) in enumerate(b.range_chunks)], dim=1).cpu()
        if reset:                                   # (the .cpu() above synchronised with the launch stream)
            for _, st in b.range_chunks:
                st[0].zero_()
                st[1].fill_(0x7f800000)
        hi = host[0].contiguous().view(torch.float32).tolist()
        lo = host[1].contiguous().view(torch.float32).tolist()
        return [(lid, (hi[i][0], lo[i][0]), (hi[i][1], lo[i][1])) for i, lid in enumerate(b.range_slots)]

    def range_quick(self):
        """The same verdict as ``range_status`` from the two-word digest the fold launch keeps
        (largest value ever split, smallest non-zero per-forward maximum): waits for the last
        forward's 8-byte copy, reads two integers.  The digest is sticky until ``range_status``
        has looked at the tables."""
        b = self.b
        if b.range is None or not b.range_slots:
            return "ok"
        if getattr(self, "_digest_fresh", False):
            b.range_sum_event.synchronize()
            vals = b.range_sum_host.tolist()
        else:
            vals = b.range_sum.tolist()               # synchronising 8-byte read
        hi, lo = (int(v) & 0xffffffff for v in vals)
        if hi > F16_MAX_BITS:                     # beyond 65504, inf or NaN bit patterns
            return "overflow"
        if getattr(self, "ignore_low", False):    # a re-calibration on 'low' changed nothing
            return "ok"
        return "low" if lo < LOW_WATER_BITS else "ok"

    def range_status(self, reset=True):
        """'ok', 'low' (some launch's largest value fell below LOW_WATER in stored units: results
        are still within bounds but the tensor should be re-calibrated), or 'overflow' (a value
        beyond the fp16 range was clamped: the results of the forwards since the last look are
        INVALID); plus the offending entries.  The tables are only walked when the digest says
        something is out of bounds."""
        if self.range_quick() == "ok":
            return "ok", []
        rep = self.range_report(reset)
        if reset and rep is not None:
            self.b.range_sum.copy_(torch.tensor([0, 0x7f800000], dtype=torch.int32))
            self.b.range_sum_host.copy_(torch.tensor([0, 0x7f800000], dtype=torch.int32))
        if rep is None:
            return "ok", []
        over = [r for r in rep if r[1][0] > F16_MAX or r[2][0] > F16_MAX or
                r[1][0] != r[1][0] or r[2][0] != r[2][0]]
        if over:
            return "overflow", over
        # per side: (largest, smallest) per-forward maximum.  smallest == +inf: no forward since
        # the last look; largest == 0: this side of the launch splits nothing (or only zeros)
        if getattr(self, "ignore_low", False):
            return "ok", []
        low = [r for r in rep if any(side[0] > 0.0 and side[1] < LOW_WATER for side in r[1:])]
        return ("low", low) if low else ("ok", [])

    def capture(self):
        """Capture the launch list in a HIP graph (launch-bound small batches)."""
        self._static_in = torch.zeros((self.b.B, self.b.input.C, self.b.H, self.b.W),
                                      device=self.b.device, dtype=torch.float32)
        self.b.input.t = self._static_in
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):  # warm-up outside capture (attribute calls, lazy loads)
                for op in self.b.ops:
                    op()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for op in self.b.ops:
                op()
        self.graph = g
        # the zero-image warm-up and capture runs above went through the range words: forget them
        # (an all-zero forward must not read as a tensor far below its calibrated range)
        torch.cuda.current_stream().synchronize()
        self.b.range_reset()
        return self


class RangeError(native.NativeError):
    """An f32s forward clamped a value although it had just been calibrated on that very input."""


class PlannedModule(torch.nn.Module):
    """Base class of the network modules: ``forward`` replays a cached Plan."""

    def describe(self, pb, x):  # pragma: no cover - interface
        raise NotImplementedError

    compute_dtype = torch.float32   # set to torch.float16 with .half_compute() (hourglass)
    f32s = None                     # None: default (on unless CN_F32S=0); see .fp32_mfma()

    def fp32_mfma(self, enable=True):
        """Compute on the plain fp32 matrix instruction (v_mfma_f32_32x32x2_f32, the round-1
        kernels) instead of f32s (three fp16 MFMAs per product): the A/B reference mode."""
        self.f32s = (not enable) if enable is not None else None
        self.drop_plans()        # exponents depend on weights and input only; packed weights are keyed by form
        return self

    def half_compute(self, enable=True):
        """fp16 activations/weights with fp32 accumulation (BASELINE configs[4]); parameters
        stay fp32 in the module, only the packed plan copies are fp16."""
        self.compute_dtype = torch.float16 if enable else torch.float32
        self.drop_plans()
        return self

    range_tracking_on = True        # see .range_tracking()

    def range_tracking(self, enable=True):
        """f32s plans with (default) or without the range words at the split sites.  Without them
        a value beyond the calibrated range is clamped SILENTLY: for measuring what the tracking
        costs (bench.py ``range_tracking_off_leg``), never for results."""
        self.range_tracking_on = bool(enable)
        self.drop_plans()
        return self

    def drop_plans(self):
        """Drop the launch lists only (mode switches): calibrated exponents and packed weights stay."""
        self.__dict__["_plans"] = {}

    max_plans = int(os.environ.get("CN_PLAN_CACHE", "8"))   # LRU bound on cached input shapes

    # ---- f32s exponents ---------------------------------------------------------------------
    def uses_f32s(self):
        split = self.f32s if self.f32s is not None else os.environ.get("CN_F32S", "1") != "0"
        return bool(split) and self.compute_dtype == torch.float32

    @property
    def exponents(self):
        """logical tensor id -> f32s exponent, or None before the first calibration."""
        return self.__dict__.get("_exps")

    def calibrate(self, x, merge=False):
        """Measure max |value| of every tensor of the network on the batch ``x`` with the plain
        fp32 kernels (no splits, nothing can saturate) and derive the f32s exponents from it:
        tensor t is then stored as real * 2^-e_t with its largest magnitude in [2^9, 2^10).
        Runs by itself on the first f32s forward, and again when a forward reports a clamped
        value (inputs far outside the calibration batch).  ``merge``: keep the larger of the old
        and the new exponent per tensor.  Costs one fp32-MFMA forward + one pass over the
        activations; the f32s plans are rebuilt (the packed weights are kept)."""
        if not x.is_cuda:
            raise native.NativeError("calibration needs a batch on a HIP device")
        B, C, H, W = x.shape
        lib = native.lib()
        with torch.no_grad():
            pb = PlanBuilder(x.device, B, H, W, dtype=torch.float32, wcache={}, calibrating=True)
            outs = self.describe(pb, pb.set_input(C))
            plan = Plan(pb, outs)
            plan.run(x.contiguous().float(), borrow=True)
            acts = [a for _, a in pb.trace if a is not None and a.lid is not None and not a.nchw]
            words = torch.zeros(len(acts) + 1, device=x.device, dtype=torch.int32)
            st = native.stream_ptr()
            xc = x.contiguous()
            native.check(lib.cn_absmax_f32(native.ptr(xc), xc.numel(), 1, 1, native.ptr(words), st),
                         "cn_absmax_f32")
            for i, a in enumerate(acts):
                wptr = ctypes.c_void_p(words.data_ptr() + 4 * (i + 1))
                native.check(lib.cn_absmax_f32(a.ptr(), a.B * a.H * a.W, a.C, a.pitch, wptr, st),
                             "cn_absmax_f32")
            vals = words.cpu().view(torch.float32).tolist()
        amax = {"input": vals[0]}
        for a, v in zip(acts, vals[1:]):
            amax[a.lid] = max(amax.get(a.lid, 0.0), v)
        exps = {lid: exponent_for(v) for lid, v in amax.items()}
        for grp in pb.groups:        # concatenated tensors share the exponent of the largest
            have = [exps[l] for l in grp if l in exps and amax.get(l, 0.0) > 0.0]
            if have:
                e = max(have)
                for l in grp:
                    exps[l] = e
        old = self.__dict__.get("_exps")
        if merge == "decay" and old:
            # a tensor drifted BELOW its calibrated range: follow it down, but by at most LOW_STEP
            # binades per re-calibration (a dark batch between bright ones must not throw the
            # bright batches' head-room away), never up past the old value from here
            for l, e in old.items():
                exps[l] = max(exps.get(l, e), e - LOW_STEP)
        elif merge and old:
            for l, e in old.items():
                exps[l] = max(e, exps.get(l, e))
        self.__dict__["_absmax"] = amax
        self.__dict__["_calibrations"] = self.__dict__.get("_calibrations", 0) + 1
        if old is not None and exps == old:
            # nothing moved (e.g. a member of a concatenation that is small next to the group's
            # largest, or a tensor that simply IS tiny): keep the plans, and stop asking
            for plan in self.__dict__.get("_plans", {}).values():
                plan.ignore_low = True
            return exps
        self.__dict__["_exps"] = exps
        self.__dict__["_plans"] = {}      # plans bake the exponents into their epilogue constants
        return exps

    def plan_for(self, B, H, W, device):
        """The plan of one input shape.  Plans (activations + launch list) are kept in an LRU of
        ``max_plans`` shapes -- --keep_res / multi-scale evaluation sees many (H, W) -- while the
        packed weights live in one per-module cache shared by all of them."""
        cache = self.__dict__.setdefault("_plans", {})
        key = (B, H, W, str(device), self.compute_dtype, self.f32s, self.range_tracking_on)
        plan = cache.pop(key, None)
        if plan is None:
            native.lib()  # raises if the HIP library is missing
            with torch.no_grad():
                pb = PlanBuilder(device, B, H, W, dtype=self.compute_dtype,
                                 wcache=self.__dict__.setdefault("_wcache", {}), split=self.f32s,
                                 exps=self.__dict__.get("_exps"),
                                 track=None if self.range_tracking_on else False)
                x = pb.set_input(3)
                outs = self.describe(pb, x)
            plan = Plan(pb, outs)
            while len(cache) >= max(1, self.max_plans):
                cache.pop(next(iter(cache)))          # least recently used
        cache[key] = plan                             # most recently used last
        return plan

    def invalidate_plans(self):
        """Drop every plan, packed weight and calibrated exponent (call after changing
        parameters in place)."""
        self.__dict__["_plans"] = {}
        self.__dict__["_wcache"] = {}
        self.__dict__["_exps"] = None

    def load_state_dict(self, *a, **kw):
        r = super().load_state_dict(*a, **kw)
        self.invalidate_plans()
        return r

    def _apply(self, fn, *a, **kw):
        # .to() / .cuda() / .cpu() / .float(): parameter storage moves, the packed copies and the
        # plans that point at them are stale
        r = super()._apply(fn, *a, **kw)
        self.invalidate_plans()
        return r

    def range_ok(self, x=None):
        """Synchronising look at the range words of every f32s plan of this module since the
        last look.  True: nothing was clamped.  False: some forward since then produced INVALID
        results; the module has been re-calibrated (on ``x`` when given) so that re-running the
        batch is valid.  A tensor that drifted far BELOW its calibrated range only schedules a
        re-calibration on the next forward (its results are still within the error bound)."""
        worst = "ok"
        for plan in list(self.__dict__.get("_plans", {}).values()):
            st, bad = plan.range_status()
            if st == "overflow":
                worst = "overflow"
                self.__dict__["_last_range_event"] = ("overflow", bad)
            elif st == "low" and worst == "ok":
                worst = "low"
                self.__dict__["_last_range_event"] = ("low", bad)
        if worst == "overflow":
            if x is not None:
                self.calibrate(x, merge=True)
            else:
                self.__dict__["_recalibrate"] = "merge"
            return False
        if worst == "low":
            self._schedule_low()
        return True

    def _schedule_low(self):
        """A forward reported a tensor far below its calibrated range (results still within the
        error bound).  Re-calibrate before the next forward -- at most once per LOW_EVERY forwards:
        inputs that alternate between far-apart scales would otherwise pay a calibration pass and
        a plan rebuild per forward."""
        nf, at = self.__dict__.get("_nfwd", 0), self.__dict__.get("_low_cal_at")
        if at is not None and nf - at < LOW_EVERY:
            if not self.__dict__.get("_low_warned"):
                self.__dict__["_low_warned"] = True
                warnings.warn("f32s: activations keep falling below the calibrated range; "
                              "re-calibration is limited to once per %d forwards" % LOW_EVERY)
            return
        self.__dict__["_low_cal_at"] = nf
        self.__dict__["_recalibrate"] = "decay"

    def forward(self, x, borrow=False, events=None, event_after=None, check=None):
        """[{head: (B,C,H/4,W/4)}] like the reference modules (fresh tensors).  ``borrow`` /
        ``events``: see Plan.run.  ``check``: f32s range check of THIS forward before it returns
        (one stream synchronisation; on a clamped value the module re-calibrates on ``x`` and
        runs again).  Default: on for the fresh-tensor form, off with ``borrow=True`` -- those
        callers (the detectors) synchronise anyway and call ``range_ok`` there."""
        if self.training:
            raise native.NativeError("centernet_amd implements the inference path only; call .eval()")
        if not x.is_cuda:
            raise native.NativeError(
                "centernet_amd runs on MI355X only (input is on %s). There is no CPU path: the "
                "CPU restatement under oracle/ is test infrastructure." % x.device)
        B, C, H, W = x.shape
        f32s = self.uses_f32s()
        if f32s:
            pending = self.__dict__.pop("_recalibrate", None)
            if self.__dict__.get("_exps") is None:
                self.calibrate(x)
            elif pending:
                self.calibrate(x, merge=True if pending == "merge" else pending)
            self.__dict__["_nfwd"] = self.__dict__.get("_nfwd", 0) + 1
        if check is None:
            check = f32s and not borrow
        plan = self.plan_for(B, H, W, x.device)
        out = plan.run(x, events=events, event_after=event_after, borrow=borrow, digest=bool(check and f32s))
        if check and f32s:
            st, bad = plan.range_status()
            if st == "overflow":
                self.calibrate(x, merge=True)
                plan = self.plan_for(B, H, W, x.device)
                if events is not None:
                    del events[:]          # the markers of the invalid forward
                out = plan.run(x, events=events, event_after=event_after, borrow=borrow)
                st, bad = plan.range_status()
                if st == "overflow":
                    raise RangeError("f32s forward still clamps after re-calibration: %r" % (bad[:3],))
            elif st == "low":
                self._schedule_low()
        return [out]
