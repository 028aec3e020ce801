"""DCNv2 operator interface (mirror of src/lib/models/networks/DCNv2/dcn_v2.py and
dcn_v2_func.py of the reference), backed by the fused HIP kernel.

``dcn_v2_forward`` is the NCHW drop-in of ``DCNv2Function(...).forward`` /
``_backend.dcn_v2_cuda_forward`` (dcn_v2_func.py:22-38).  The ``DCN`` module keeps
the reference's parameter names (``weight``, ``bias``, ``conv_offset_mask.*``).
Inference only: there is no backward.
"""
import math

import torch
from torch import nn
from torch.nn.modules.utils import _pair

from . import native


def _infer_shape(input, weight, stride, padding, dilation):
    # dcn_v2_func.py:64-73
    n = input.size(0)
    channels_out = weight.size(0)
    height, width = input.shape[2:4]
    kernel_h, kernel_w = weight.shape[2:4]
    height_out = (height + 2 * padding - (dilation * (kernel_h - 1) + 1)) // stride + 1
    width_out = (width + 2 * padding - (dilation * (kernel_w - 1) + 1)) // stride + 1
    return (n, channels_out, height_out, width_out)


def dcn_v2_forward(input, offset, mask, weight, bias, stride=1, padding=1, dilation=1,
                   deformable_groups=1, apply_mask_sigmoid=False):
    """input (B,Cin,H,W), offset (B,2*9*dg,Ho,Wo), mask (B,9*dg,Ho,Wo) -> (B,Cout,Ho,Wo)."""
    if not input.is_cuda:
        raise NotImplementedError  # same as the reference (dcn_v2_func.py:23-24)
    lib = native.lib()
    input, offset, mask = input.contiguous(), offset.contiguous(), mask.contiguous()
    native.require_f32(input, offset, mask, weight, bias)
    weight = weight.detach().to(input.device).contiguous()   # a module left on the host still runs
    bias = bias.detach().to(input.device).contiguous()
    B, Cin, H, W = input.shape
    Cout, Cin_w, kh, kw = weight.shape
    if Cin_w != Cin:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (Cin, Cin_w))
    shape = _infer_shape(input, weight, stride, padding, dilation)
    if tuple(offset.shape) != (B, deformable_groups * 2 * kh * kw, shape[2], shape[3]) or \
            tuple(mask.shape) != (B, deformable_groups * kh * kw, shape[2], shape[3]):
        raise RuntimeError("offset/mask shape does not match the output grid")
    output = torch.empty(shape, device=input.device, dtype=torch.float32)
    ws_bytes = lib.cn_dcn_v2_forward_workspace_bytes(B, Cin, H, W, Cout, kh, kw, native.LAYOUT_NCHW)
    ws = torch.empty(max(ws_bytes, 16), device=input.device, dtype=torch.uint8)
    rc = lib.cn_dcn_v2_forward_f32(native.ptr(input), native.ptr(weight), native.ptr(bias),
                                   native.ptr(offset), native.ptr(mask), native.ptr(output),
                                   B, Cin, H, W, Cout, kh, kw, stride, stride, padding, padding,
                                   dilation, dilation, deformable_groups, int(apply_mask_sigmoid),
                                   native.ptr(ws), ws_bytes, native.stream_ptr())
    native.check(rc, "cn_dcn_v2_forward_f32")
    return output


class DCNv2(nn.Module):
    """dcn_v2.py:14-41 (parameters + explicit offset/mask inputs)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1,
                 deformable_groups=1):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1.0 / math.sqrt(n)
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
            self.bias.zero_()

    def forward(self, input, offset, mask):
        return dcn_v2_forward(input, offset, mask, self.weight, self.bias, self.stride,
                              self.padding, self.dilation, self.deformable_groups)


class DCN(DCNv2):
    """dcn_v2.py:44-70: offsets and mask come from ``conv_offset_mask`` (zero-init)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1,
                 deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation,
                         deformable_groups)
        self.conv_offset_mask = nn.Conv2d(
            self.in_channels,
            self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1],
            kernel_size=self.kernel_size, stride=(self.stride, self.stride),
            padding=(self.padding, self.padding), bias=True)
        with torch.no_grad():
            self.conv_offset_mask.weight.zero_()
            self.conv_offset_mask.bias.zero_()

    def _tuned(self, cin):
        """The configuration CenterNet instantiates (resnet_dcn.py:221-223, pose_dla_dcn.py:352):
        served by the NHWC MFMA kernel; everything else by the general-domain kernel."""
        return (tuple(self.kernel_size) == (3, 3) and self.stride == 1 and self.padding == 1 and
                self.dilation == 1 and self.deformable_groups == 1 and cin % 4 == 0)

    def forward(self, input):
        """Stand-alone use (NCHW in/out).  Inside a network the plan fuses this with the
        following BatchNorm+ReLU and stays in NHWC (engine.PlanBuilder.dcn)."""
        from .engine import PlanBuilder, Act, exponent_for
        from . import native
        if not input.is_cuda:
            raise NotImplementedError
        B, C, H, W = input.shape
        # f32s arithmetic splits x * 2^-e: e from this very input (one reduction + a host read;
        # the stand-alone form builds its launch list per call anyway)
        xin = input.contiguous().float()
        word = torch.zeros(1, device=input.device, dtype=torch.int32)
        native.check(native.lib().cn_absmax_f32(native.ptr(xin), xin.numel(), 1, 1, native.ptr(word),
                                                native.stream_ptr()), "cn_absmax_f32")
        ex = exponent_for(float(word.cpu().view(torch.float32)[0]))
        pb = PlanBuilder(input.device, B, H, W, exps={"x": ex})
        if self._tuned(C):
            x_nhwc = input.permute(0, 2, 3, 1).contiguous()
            y = pb.dcn(Act(x_nhwc, B, H, W, C, exp=pb._exp("x"), lid="x"), self, out_plain=True)
            for op in pb.ops:
                op()
            return y.t.permute(0, 3, 1, 2).contiguous()
        # general domain (dcn_v2.py:64-70 literally): conv_offset_mask as one implicit-GEMM launch
        # (channels zero-padded to the kernel's 4-channel granule), then chunk / cat / sigmoid
        # folded into the operator call (the first 2/3 of the channels ARE cat(o1, o2)).
        cp = (C + 3) // 4 * 4
        x_nhwc = input.new_zeros((B, H, W, cp))
        x_nhwc[..., :C] = input.permute(0, 2, 3, 1)
        com = self.conv_offset_mask
        w = com.weight.detach().new_zeros((com.weight.shape[0], cp) + tuple(com.weight.shape[2:]))
        w[:, :C] = com.weight.detach()
        om = pb.conv(Act(x_nhwc, B, H, W, cp, exp=pb._exp("x"), lid="x"), w, bias=com.bias,
                     stride=self.stride, padding=self.padding, out_nchw=True)
        for op in pb.ops:
            op()
        n_off = 2 * self.deformable_groups * self.kernel_size[0] * self.kernel_size[1]
        return dcn_v2_forward(input, om.t[:, :n_off], om.t[:, n_off:], self.weight, self.bias,
                              self.stride, self.padding, self.dilation, self.deformable_groups,
                              apply_mask_sigmoid=True)
