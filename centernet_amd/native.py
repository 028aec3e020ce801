"""ctypes binding of libcenternet_amd.so -- the C ABI declared in include/centernet_amd.h.

There is no fallback: if the HIP library has not been built (``python -c
"import __graft_entry__ as g; g.build()"`` or ``make -C centernet_amd/csrc``) every
entry point raises.  PyTorch is used only for device memory and streams; tensors
cross this boundary as raw device pointers.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CENTERNET_AMD_LIB: another build of the same library (A/B measurements, tools/build_variant.sh)
LIB_PATH = os.environ.get("CENTERNET_AMD_LIB") or os.path.join(_HERE, "libcenternet_amd.so")
CSRC = os.path.join(_HERE, "csrc")

CN_OK = 0
LAYOUT_NCHW = 0
LAYOUT_NHWC = 1
DTYPE_F32 = 0
DTYPE_F16 = 1
DTYPE_F32S = 2        # fp32 values as fp16 (high, low) pairs: three fp16 MFMAs per product
CONV_X_PLAIN, CONV_Y_PLAIN, CONV_R_PLAIN, CONV_STEM_F32S, CONV_STEM_MAXPOOL = 1, 2, 4, 8, 16
CONV_STEM_Y_F32S = 32

_lib = None


class NativeError(RuntimeError):
    pass


class F32sCtl(ctypes.Structure):
    """Mirror of ``cn_f32s_ctl``: range control of the f32s kernels (multipliers that carry the
    per-tensor exponents + the device words that receive the largest |value| a launch split)."""
    _fields_ = [("x_mul", ctypes.c_float), ("res_mul", ctypes.c_float), ("range", ctypes.c_void_p)]


class ConvDesc(ctypes.Structure):
    """Mirror of ``cn_conv_desc`` (include/centernet_amd.h)."""
    _fields_ = [(n, ctypes.c_int) for n in (
        "B", "H", "W", "Cin", "Ho", "Wo", "Cout", "KH", "KW", "stride", "pad_h", "pad_w",
        "dil", "in_layout", "in_pitch", "out_layout", "out_pitch", "OH", "OW", "oy_mul",
        "oy_add", "ox_mul", "ox_add", "relu", "dtype", "flags", "res_pitch")] + [("ctl", F32sCtl)]


class HeadOut(ctypes.Structure):
    """Mirror of ``cn_head_out`` (include/centernet_amd.h)."""
    _fields_ = [("w", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("cout", ctypes.c_int), ("reserved", ctypes.c_int), ("oscale", ctypes.c_void_p),
                ("w_frag", ctypes.c_void_p)]


def build(force=False, verbose=False):
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j8"]
    if force:
        args.append("-B")
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(args, stdout=out)
    if not os.path.exists(LIB_PATH):
        raise NativeError("build did not produce %s" % LIB_PATH)
    return LIB_PATH


def _declare(lib):
    vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    lib.cn_version.restype = i
    lib.cn_status_string.restype = ctypes.c_char_p
    lib.cn_status_string.argtypes = [i]
    lib.cn_arch.restype = ctypes.c_char_p
    lib.cn_set_tuning.restype = i
    lib.cn_set_tuning.argtypes = [i, i]
    lib.cn_dcn_v2_forward_workspace_bytes.restype = sz
    lib.cn_dcn_v2_forward_workspace_bytes.argtypes = [i] * 8
    lib.cn_dcn_v2_forward_f32.restype = i
    lib.cn_dcn_v2_forward_f32.argtypes = [vp] * 6 + [i] * 15 + [vp, sz, vp]
    lib.cn_dcn_v2_forward_nhwc_f32.restype = i
    lib.cn_dcn_v2_forward_nhwc_f32.argtypes = [vp, vp, vp, vp, i, vp, vp, vp] + [i] * 7 + [vp, sz, vp]
    lib.cn_dcn_v2_forward_nhwc_workspace_bytes.restype = sz
    lib.cn_dcn_v2_forward_nhwc_workspace_bytes.argtypes = [i] * 5
    lib.cn_f32_to_f32s.restype = i
    lib.cn_f32_to_f32s.argtypes = [vp, vp, sz, i, i, i, vp]
    lib.cn_f32s_to_f32.restype = i
    lib.cn_f32s_to_f32.argtypes = [vp, vp, sz, i, i, i, vp]
    f = ctypes.c_float
    ctl = ctypes.POINTER(F32sCtl)
    lib.cn_f32_to_f32s_scaled.restype = i
    lib.cn_f32_to_f32s_scaled.argtypes = [vp, vp, sz, i, i, i, f, vp, vp]
    lib.cn_f32s_to_f32_scaled.restype = i
    lib.cn_f32s_to_f32_scaled.argtypes = [vp, vp, sz, i, i, i, f, vp]
    lib.cn_absmax_f32.restype = i
    lib.cn_absmax_f32.argtypes = [vp, sz, i, i, vp, vp]
    lib.cn_range_fold.restype = i
    lib.cn_range_fold.argtypes = [vp, vp, vp, i, vp]
    lib.cn_range_fold_digest.restype = i
    lib.cn_range_fold_digest.argtypes = [vp, vp, vp, vp, i, vp]
    lib.cn_packed_head_w2_bytes.restype = sz
    lib.cn_packed_head_w2_bytes.argtypes = [i]
    lib.cn_pack_head_w2_f32s.restype = i
    lib.cn_pack_head_w2_f32s.argtypes = [vp, vp, i, vp]
    lib.cn_flip_average_f32.restype = i
    lib.cn_flip_average_f32.argtypes = [vp, vp, i, i, i, vp, vp, i, vp]
    lib.cn_calib_mfma_f16.restype = ctypes.c_double
    lib.cn_calib_mfma_f16.argtypes = [vp, i, vp]
    lib.cn_calib_copy.restype = i
    lib.cn_calib_copy.argtypes = [vp, vp, sz, vp]
    lib.cn_calib_clock.restype = i
    lib.cn_calib_clock.argtypes = [vp, i, vp]
    lib.cn_calib_latency.restype = i
    lib.cn_calib_latency.argtypes = [vp, ctypes.c_uint32, i, vp, vp, vp]
    lib.cn_maxpool_nhwc_f32s.restype = i
    lib.cn_maxpool_nhwc_f32s.argtypes = [vp, vp, i, i, i, i, i, i, i, i, i, f, vp, vp]
    lib.cn_maxpool_nhwc_scaled.restype = i
    lib.cn_maxpool_nhwc_scaled.argtypes = [vp, vp, i, i, i, i, i, i, i, i, f, vp]
    lib.cn_dcn_v2_forward_nhwc.restype = i
    lib.cn_dcn_v2_forward_nhwc.argtypes = [vp, vp, vp, vp, i, vp, vp, vp] + [i] * 10 + [ctl, vp, sz, vp]
    lib.cn_pack_deconv4x4s2_weight.restype = i
    lib.cn_pack_deconv4x4s2_weight.argtypes = [vp, vp, i, i, i, vp]
    lib.cn_conv_transpose4x4s2.restype = i
    lib.cn_conv_transpose4x4s2.argtypes = [vp, vp, vp, vp, vp] + [i] * 10 + [ctl, vp]
    lib.cn_heads3x3_1x1.restype = i
    lib.cn_heads3x3_1x1.argtypes = [vp, i, i, i, i, i, vp, vp, vp, i, i, vp, i, i, ctl, vp]
    lib.cn_packed_conv_weight_floats.restype = sz
    lib.cn_packed_conv_weight_floats.argtypes = [i] * 4
    lib.cn_pack_conv_weight_f32.restype = i
    lib.cn_pack_conv_weight_f32.argtypes = [vp, vp, i, i, i, i, vp]
    lib.cn_stem_maxpool_supported.restype = i
    lib.cn_stem_maxpool_supported.argtypes = [ctypes.POINTER(ConvDesc)]
    lib.cn_stem_f32s_supported.restype = i
    lib.cn_stem_f32s_supported.argtypes = [ctypes.POINTER(ConvDesc)]
    lib.cn_conv2d_f32.restype = i
    lib.cn_conv2d_f32.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp]
    lib.cn_packed_conv_weight_elems.restype = sz
    lib.cn_packed_conv_weight_elems.argtypes = [i] * 5
    lib.cn_pack_conv_weight.restype = i
    lib.cn_pack_conv_weight.argtypes = [vp, vp, i, i, i, i, i, vp]
    lib.cn_conv2d_res_pitch_supported.restype = i
    lib.cn_conv2d_res_pitch_supported.argtypes = [ctypes.POINTER(ConvDesc)]
    lib.cn_conv2d_workspace_bytes.restype = sz
    lib.cn_conv2d_workspace_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
    lib.cn_conv2d.restype = i
    lib.cn_conv2d.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.cn_upsample2x_add_f16.restype = i
    lib.cn_upsample2x_add_f16.argtypes = [vp, vp, vp, i, i, i, i, vp]
    lib.cn_packed_deconv4x4s2_weight_floats.restype = sz
    lib.cn_packed_deconv4x4s2_weight_floats.argtypes = [i, i]
    lib.cn_pack_deconv4x4s2_weight_f32.restype = i
    lib.cn_pack_deconv4x4s2_weight_f32.argtypes = [vp, vp, i, i, vp]
    lib.cn_conv_transpose4x4s2_f32.restype = i
    lib.cn_conv_transpose4x4s2_f32.argtypes = [vp] * 5 + [i] * 8 + [vp]
    lib.cn_maxpool3x3s2_nhwc_f32.restype = i
    lib.cn_maxpool3x3s2_nhwc_f32.argtypes = [vp, vp, i, i, i, i, vp]
    lib.cn_maxpool_nhwc_f32.restype = i
    lib.cn_maxpool_nhwc_f32.argtypes = [vp, vp, i, i, i, i, i, i, i, vp]
    lib.cn_maxpool_nhwc.restype = i
    lib.cn_maxpool_nhwc.argtypes = [vp, vp, i, i, i, i, i, i, i, i, vp]
    lib.cn_dw_conv_transpose_f32.restype = i
    lib.cn_dw_conv_transpose_f32.argtypes = [vp, vp, vp, vp, i, i, i, i, i, vp]
    lib.cn_copy_channels_f32.restype = i
    lib.cn_copy_channels_f32.argtypes = [vp, i, vp, i, sz, i, vp]
    lib.cn_upsample2x_add_f32.restype = i
    lib.cn_upsample2x_add_f32.argtypes = [vp, vp, vp, i, i, i, i, vp]
    lib.cn_heads3x3_1x1_f32.restype = i
    lib.cn_heads3x3_1x1_f32.argtypes = [vp, i, i, i, i, i, vp, vp, i, i, ctypes.POINTER(HeadOut), vp]
    lib.cn_warp_normalize_u8_f32.restype = i
    lib.cn_warp_normalize_u8_f32.argtypes = [vp, i, i, i, ctypes.POINTER(ctypes.c_double), i, i,
                                             ctypes.POINTER(ctypes.c_float),
                                             ctypes.POINTER(ctypes.c_float), i, vp, vp]
    lib.cn_warp_normalize_u8_f32_batch.restype = i
    lib.cn_warp_normalize_u8_f32_batch.argtypes = [vp, i, sz, i, i, i, ctypes.POINTER(ctypes.c_double), i, i,
                                                   ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                                   i, vp, vp]
    lib.cn_ctdet_post_process_f32.restype = i
    lib.cn_ctdet_post_process_f32.argtypes = [vp, i, i, i, vp, i, ctypes.c_float, vp, vp, vp]
    lib.cn_resize_bilinear_u8.restype = i
    lib.cn_resize_bilinear_u8.argtypes = [vp, i, i, i, i, i, vp, vp]
    lib.cn_warp_affine_u8_host.restype = i
    lib.cn_warp_affine_u8_host.argtypes = [vp, i, i, i, ctypes.POINTER(ctypes.c_double), i, i, vp]
    lib.cn_resize_linear_u8_host.restype = i
    lib.cn_resize_linear_u8_host.argtypes = [vp, i, i, i, i, i, vp]
    lib.cn_normalize_u8_chw_f32_host.restype = i
    lib.cn_normalize_u8_chw_f32_host.argtypes = [vp, i, i, vp, vp, vp]
    lib.cn_soft_nms_f32.restype = i
    lib.cn_soft_nms_f32.argtypes = [vp, i, i, ctypes.c_float, ctypes.c_float, ctypes.c_float, i]
    lib.cn_nchw_to_nhwc_f32.restype = i
    lib.cn_nchw_to_nhwc_f32.argtypes = [vp, vp, i, i, i, i, i, vp]
    lib.cn_nhwc_to_nchw_f32.restype = i
    lib.cn_nhwc_to_nchw_f32.argtypes = [vp, vp, i, i, i, i, i, vp]
    lib.cn_ctdet_decode_workspace_bytes.restype = sz
    lib.cn_ctdet_decode_workspace_bytes.argtypes = [i] * 5
    lib.cn_decode_state_region.restype = i
    lib.cn_decode_state_region.argtypes = [i] * 5 + [ctypes.POINTER(ctypes.c_size_t)] * 2
    lib.cn_ctdet_decode_f32.restype = i
    lib.cn_ctdet_decode_f32.argtypes = [vp, vp, vp] + [i] * 7 + [vp, vp, vp, sz, vp]
    lib.cn_nms_topk_channel_f32.restype = i
    lib.cn_nms_topk_channel_f32.argtypes = [vp] + [i] * 6 + [vp, vp, vp, sz, vp]
    lib.cn_topk_f32.restype = i
    lib.cn_topk_f32.argtypes = [vp] + [i] * 6 + [vp, vp, vp, vp, sz, vp]
    lib.cn_gather_feat_f32.restype = i
    lib.cn_gather_feat_f32.argtypes = [vp, vp, vp] + [i] * 5 + [vp]
    lib.cn_ddd_decode_workspace_bytes.restype = sz
    lib.cn_ddd_decode_workspace_bytes.argtypes = [i] * 5
    lib.cn_ddd_decode_f32.restype = i
    lib.cn_ddd_decode_f32.argtypes = [vp] * 6 + [i] * 6 + [vp, vp, sz, vp]
    lib.cn_exct_decode_workspace_bytes.restype = sz
    lib.cn_exct_decode_workspace_bytes.argtypes = [i] * 5
    lib.cn_exct_aggregate_f32.restype = i
    lib.cn_exct_aggregate_f32.argtypes = [vp, vp, i, i, i, i, i, ctypes.c_float, vp]
    lib.cn_exct_decode_f32.restype = i
    lib.cn_exct_decode_f32.argtypes = [vp] * 9 + [i] * 5 + [ctypes.c_float, ctypes.c_float, i, i,
                                                        vp, vp, sz, vp]
    lib.cn_agnex_ct_decode_workspace_bytes.restype = sz
    lib.cn_agnex_ct_decode_workspace_bytes.argtypes = [i] * 5
    lib.cn_agnex_ct_decode_f32.restype = i
    lib.cn_agnex_ct_decode_f32.argtypes = [vp] * 9 + [i] * 5 + [ctypes.c_float, ctypes.c_float, i, i,
                                                            vp, vp, sz, vp]
    lib.cn_multi_pose_decode_workspace_bytes.restype = sz
    lib.cn_multi_pose_decode_workspace_bytes.argtypes = [i] * 6
    lib.cn_multi_pose_decode_f32.restype = i
    lib.cn_multi_pose_decode_f32.argtypes = [vp] * 6 + [i] * 7 + [vp, vp, sz, vp]


def lib():
    """Load the HIP library; raise loudly if it is missing (no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                "%s is missing: build it with `make -C %s` (hipcc, gfx950). "
                "centernet_amd has no CPU or PyTorch fallback." % (LIB_PATH, CSRC))
        l = ctypes.CDLL(LIB_PATH)
        _declare(l)
        _lib = l
    return _lib


def check(rc, what):
    if rc != CN_OK:
        raise NativeError("%s failed: %s (%d)" % (what, lib().cn_status_string(rc).decode(), rc))


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Raw device pointer of a CUDA/HIP fp32/int32 tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NativeError("centernet_amd kernels need tensors on a HIP device (got %s); "
                          "there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise NativeError("tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def require_f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise NativeError("fp32 tensors required, got %s" % t.dtype)
