"""f32s (fp16 high/low pairs, three fp16 MFMAs per product) against the fp32-MFMA kernels:
speed and accuracy of single conv launches at the resdcn_18 / dla_34 trunk shapes, B = 32.
GPU box only.  Usage: python tools/bench_f32s.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from centernet_amd import native
from centernet_amd.native import ConvDesc, LAYOUT_NHWC, DTYPE_F32, DTYPE_F32S

dev = torch.device("cuda:0")
lib = native.lib()
B = int(os.environ.get("B", 32))
SHAPES = [(64, 128, 128, 64), (128, 64, 64, 128), (256, 32, 32, 256), (512, 16, 16, 512),
          (64, 128, 128, 192), (128, 64, 64, 27)]


def prescale(w):
    """Per-output-channel power of two that brings max|w| into [0.5, 1): exact, undone in `scale`."""
    m = w.abs().amax(dim=(1, 2, 3)).clamp_min(1e-30)
    e = torch.frexp(m)[1]                      # m = mant * 2^e, mant in [0.5, 1)
    return torch.ldexp(w, (-e).view(-1, 1, 1, 1).expand_as(w)), torch.ldexp(torch.ones_like(m), e)


def run(ci, H, W, co, dtype, x_plain, w, scale_in=None, residual=None, iters=20):
    pitch_o = (co + 31) // 32 * 32
    st = native.stream_ptr()
    if dtype == DTYPE_F32S:
        ws, sc = prescale(w)
        x = torch.empty((B, H, W, ci), device=dev)
        native.check(lib.cn_f32_to_f32s(native.ptr(x_plain), native.ptr(x), B * H * W, ci, ci, ci, st), "cvt")
    else:
        ws, sc, x = w, None, x_plain
    n = lib.cn_packed_conv_weight_elems(co, ci, 3, 3, dtype)
    wp = torch.empty(n, device=dev)
    native.check(lib.cn_pack_conv_weight(native.ptr(ws.to(dev).contiguous()), native.ptr(wp), co, ci, 3, 3, dtype, st), "pack")
    y = torch.zeros((B, H, W, pitch_o), device=dev)
    d = ConvDesc(B=B, H=H, W=W, Cin=ci, Ho=H, Wo=W, Cout=co, KH=3, KW=3, stride=1, pad_h=1, pad_w=1,
                 dil=1, in_layout=LAYOUT_NHWC, in_pitch=ci, out_layout=LAYOUT_NHWC, out_pitch=pitch_o,
                 OH=H, OW=W, oy_mul=1, oy_add=0, ox_mul=1, ox_add=0, relu=0, dtype=dtype)
    scp = native.ptr(sc.to(dev)) if sc is not None else None

    def launch():
        rc = lib.cn_conv2d(ctypes.byref(d), native.ptr(x), native.ptr(wp), scp, None, None, native.ptr(y),
                           None, 0, st)
        assert rc == 0, rc
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        launch()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    if dtype == DTYPE_F32S:
        out = torch.empty_like(y)
        native.check(lib.cn_f32s_to_f32(native.ptr(y), native.ptr(out), B * H * W, co, pitch_o, pitch_o, st), "cvt")
        y = out
    return ms, y[..., :co]


print("%-26s %22s %22s   max|f32s-ref| max|f32-ref| (ref = fp64 conv, rel. to rms)" % ("shape (Cin,H,W,Cout)", "fp32 MFMA", "f32s"))
for (ci, H, W, co) in SHAPES:
    g = torch.Generator().manual_seed(ci + co)
    x = torch.randn((B, H, W, ci), generator=g).relu_().to(dev)
    w = torch.randn((co, ci, 3, 3), generator=g) * (2.0 / (ci * 9)) ** 0.5
    fl = 2.0 * B * H * W * co * ci * 9
    ms0, y0 = run(ci, H, W, co, DTYPE_F32, x, w)
    lib.cn_set_tuning(20, 0)
    ms2, _ = run(ci, H, W, co, DTYPE_F32S, x, w)
    lib.cn_set_tuning(20, 1)
    ms1, y1 = run(ci, H, W, co, DTYPE_F32S, x, w)
    nb = min(B, 2)
    ref = torch.nn.functional.conv2d(x[:nb].permute(0, 3, 1, 2).double().cpu(), w.double(), padding=1).permute(0, 2, 3, 1)
    rms = float(ref.pow(2).mean().sqrt())
    e0 = float((y0[:nb].double().cpu() - ref).abs().max()) / rms
    e1 = float((y1[:nb].double().cpu() - ref).abs().max()) / rms
    print("%-26s %7.3f ms %7.1f TF   %7.3f ms %7.1f TF   %.2e  %.2e | register-streamed weights (key 20=0) %7.3f ms %6.1f TF" % (
        str((ci, H, W, co)), ms0, fl / ms0 / 1e9, ms1, fl / ms1 / 1e9, e1, e0, ms2, fl / ms2 / 1e9))
