"""Error behaviour of the C ABI on a device: return codes (never a longjmp / exception from
native code, never a silent fallback), mirrored from the reference's THError / torch checks:
  dcn_v2_cuda.c:20-38 (contiguity, shapes), decode.py:106 (topk k out of range)."""
import ctypes

import numpy as np
import pytest
import torch

from centernet_amd import native

pytestmark = pytest.mark.gpu
OK, SHAPE, UNSUP, WS, LAUNCH, NULL, ALIGN = 0, -1, -2, -3, -4, -5, -6


def test_decode_error_codes(dev):
    lib = native.lib()
    B, C, H, W, K = 1, 3, 16, 16, 10
    heat = torch.rand((B, C, H, W), device=dev)
    wh = torch.rand((B, 2, H, W), device=dev)
    dets = torch.empty((B, K, 6), device=dev)
    need = lib.cn_ctdet_decode_workspace_bytes(B, C, H, W, K)
    ws = torch.empty(need, device=dev, dtype=torch.uint8)
    st = native.stream_ptr()
    p = native.ptr

    def call(heat_=heat, wh_=wh, dets_=dets, ws_=ws, nbytes=need, k=K, b=B):
        return lib.cn_ctdet_decode_f32(p(heat_), p(wh_), None, b, C, H, W, k, 0, 0, p(dets_), None,
                                       p(ws_), nbytes, st)
    assert call() == OK
    assert call(nbytes=need - 1) == WS
    assert call(heat_=None) == NULL and call(dets_=None) == NULL and call(ws_=None) == NULL
    assert call(k=H * W + 1) == SHAPE          # torch.topk: k out of range
    assert call(k=129) == UNSUP                # K > 128 is not built
    assert call(b=0) == SHAPE
    assert lib.cn_ctdet_decode_workspace_bytes(B, C, H, W, 0) == 0
    mis = torch.rand(B * C * H * W + 1, device=dev)[1:].view(B, C, H, W)   # 4-byte aligned only
    assert call(heat_=mis) == ALIGN
    torch.cuda.synchronize()
    assert lib.cn_status_string(WS).decode() and lib.cn_status_string(ALIGN).decode()


def test_dcn_error_codes(dev):
    lib = native.lib()
    B, Cin, H, W, Cout = 1, 8, 6, 6, 8
    x = torch.rand((B, Cin, H, W), device=dev)
    w = torch.rand((Cout, Cin, 3, 3), device=dev)
    bias = torch.rand(Cout, device=dev)
    off = torch.zeros((B, 18, H, W), device=dev)
    mask = torch.ones((B, 9, H, W), device=dev)
    y = torch.empty((B, Cout, H, W), device=dev)
    need = lib.cn_dcn_v2_forward_workspace_bytes(B, Cin, H, W, Cout, 3, 3, native.LAYOUT_NCHW)
    ws = torch.empty(need, device=dev, dtype=torch.uint8)
    p, st = native.ptr, native.stream_ptr()

    def call(k=3, stride=1, pad=1, dil=1, dg=1, nbytes=need, x_=x, cin=Cin):
        return lib.cn_dcn_v2_forward_f32(p(x_), p(w), p(bias), p(off), p(mask), p(y), B, cin, H, W,
                                         Cout, k, k, stride, stride, pad, pad, dil, dil, dg, 0,
                                         p(ws), nbytes, st)
    assert call() == OK
    assert call(nbytes=need - 1) == WS
    assert call(x_=None) == NULL
    # invalid geometry (the reference raises THError, dcn_v2_cuda.c:33-38): return codes
    assert call(stride=0) == SHAPE and call(dil=0) == SHAPE and call(dg=0) == SHAPE
    assert call(dg=3) == SHAPE                      # Cin % deformable_group != 0
    assert call(pad=-1) == SHAPE and call(k=0) == SHAPE
    assert call(cin=0) == SHAPE
    assert call(k=9, pad=0) == SHAPE                # kernel larger than the padded map: Ho <= 0
    torch.cuda.synchronize()


def test_conv_and_pre_process_error_codes(dev):
    lib = native.lib()
    d = native.ConvDesc(B=1, H=8, W=8, Cin=8, Ho=8, Wo=8, Cout=8, KH=3, KW=3, stride=1, pad_h=1,
                        pad_w=1, dil=1, in_layout=native.LAYOUT_NHWC, in_pitch=8,
                        out_layout=native.LAYOUT_NHWC, out_pitch=8, OH=8, OW=8, oy_mul=1, oy_add=0,
                        ox_mul=1, ox_add=0, relu=0, dtype=native.DTYPE_F32)
    x = torch.rand((1, 8, 8, 8), device=dev)
    wp = torch.zeros(lib.cn_packed_conv_weight_floats(8, 8, 3, 3), device=dev)
    y = torch.empty((1, 8, 8, 8), device=dev)
    p, st = native.ptr, native.stream_ptr()
    assert lib.cn_conv2d_f32(ctypes.byref(d), p(x), p(wp), None, None, None, p(y), st) == OK
    assert lib.cn_conv2d_f32(ctypes.byref(d), None, p(wp), None, None, None, p(y), st) == NULL
    bad = native.ConvDesc.from_buffer_copy(d)
    bad.Ho = 7                                    # inconsistent output size
    assert lib.cn_conv2d_f32(ctypes.byref(bad), p(x), p(wp), None, None, None, p(y), st) == SHAPE
    bad = native.ConvDesc.from_buffer_copy(d)
    bad.in_pitch = 6                              # pitch smaller than Cin
    assert lib.cn_conv2d_f32(ctypes.byref(bad), p(x), p(wp), None, None, None, p(y), st) != OK
    # pre-process: pitch too small, zero std
    img = torch.zeros((4, 4, 3), device=dev, dtype=torch.uint8)
    out = torch.empty((1, 3, 4, 4), device=dev)
    m = (ctypes.c_double * 6)(1, 0, 0, 0, 1, 0)
    mean, std, zero = (ctypes.c_float * 3)(0, 0, 0), (ctypes.c_float * 3)(1, 1, 1), (ctypes.c_float * 3)(1, 0, 1)
    assert lib.cn_warp_normalize_u8_f32(p(img), 4, 4, 12, m, 4, 4, mean, std, 0, p(out), st) == OK
    assert lib.cn_warp_normalize_u8_f32(p(img), 4, 4, 11, m, 4, 4, mean, std, 0, p(out), st) == SHAPE
    assert lib.cn_warp_normalize_u8_f32(p(img), 4, 4, 12, m, 4, 4, mean, zero, 0, p(out), st) == SHAPE
    assert lib.cn_warp_normalize_u8_f32(None, 4, 4, 12, m, 4, 4, mean, std, 0, p(out), st) == NULL
    torch.cuda.synchronize()


def test_python_layer_raises_on_cpu_tensors():
    from centernet_amd.decode import ctdet_decode
    with pytest.raises(native.NativeError):
        ctdet_decode(torch.rand(1, 2, 8, 8), torch.rand(1, 2, 8, 8))


def test_box_calibration_probes(dev):
    """The measurement aids of bench.py's box_calibration / clock_under_load (no reference counterpart):
    the dependent-load walk and the device-scope atomic chain report plausible times and follow the
    chain they are given; the clock probe reads a core clock of a few GHz; error codes for bad arguments."""
    lib = native.lib()
    st, p = native.stream_ptr(), native.ptr
    n, stride, steps = 4096, 5 * 32, 512
    chain = (torch.arange(n, device=dev, dtype=torch.int32) + stride) % n
    atom = torch.zeros(64, device=dev, dtype=torch.int32)
    out = torch.zeros(4, device=dev, dtype=torch.int64)
    assert lib.cn_calib_latency(p(chain), 0, steps, p(atom), p(out), st) == OK
    torch.cuda.synchronize()
    t_load, t_atom, sink = out.cpu().tolist()[:3]
    assert 0 < t_load * 10.0 / steps < 5000 and 0 < t_atom * 10.0 / steps < 20000      # ns per step
    assert int(atom[0]) == steps                                  # one addition of 1 per step
    # the walk ends where `steps` hops of the chain lead (the sink adds the last fetched atomic value)
    assert sink == (steps * stride) % n + (steps - 1)
    assert lib.cn_calib_latency(None, 0, steps, p(atom), p(out), st) == NULL
    assert lib.cn_calib_latency(p(chain), 0, 0, p(atom), p(out), st) == SHAPE
    assert lib.cn_calib_clock(p(out), 2000, st) == OK
    torch.cuda.synchronize()
    cyc, ticks = out.cpu().tolist()[:2]
    assert ticks >= 2000 and 300.0 < 100.0 * cyc / ticks < 5000.0     # MHz
    assert lib.cn_calib_clock(None, 2000, st) == NULL and lib.cn_calib_clock(p(out), 0, st) == SHAPE
