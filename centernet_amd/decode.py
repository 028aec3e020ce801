"""Heat-map decode entry points (mirror of src/lib/models/decode.py), backed by the
fused HIP kernels in csrc/cn_decode.hip / cn_pose.hip.

Same callables, same argument meaning, tensors in / tensor out:
    ctdet_decode(heat, wh, reg=None, cat_spec_wh=False, K=100)      decode.py:464-495
    multi_pose_decode(heat, wh, kps, reg, hm_hp, hp_offset, K)      decode.py:497-571
    _nms / _topk / _topk_channel                                    decode.py:9-15, 92-119
``heat`` is post-sigmoid as in the reference; pass ``apply_sigmoid=True`` with logits
to fuse ``hm.sigmoid_()`` (detectors/ctdet.py:31) into the same pass over the heat-map.
"""
import torch

from . import native

_ws_cache = {}


def _workspace(nbytes, device):
    key = str(device)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 256), device=device, dtype=torch.uint8)
        _ws_cache[key] = ws
    return ws


_STATE_CLEAN = 4096    # CN_DECODE_STATE_CLEAN: the workspace's image state words are zero (see below)
_own_ws = {}
_verify_state = __import__("os").environ.get("CN_DECODE_VERIFY_STATE") == "1"


def _own_workspace(entry, nbytes, device, shape):
    """A workspace that only `entry` with this shape on this stream ever touches, zeroed once: the
    one-launch image-level decode keeps three state words per image in it, needs them zero on entry
    and leaves them zero on exit (include/centernet_amd.h, CN_DECODE_STATE_CLEAN) -- so the call is
    ONE kernel launch, without the fill the library otherwise puts in front."""
    key = (str(device), entry, torch.cuda.current_stream(device).cuda_stream) + tuple(shape)
    ws = _own_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        if len(_own_ws) >= 32:
            _own_ws.clear()
        ws = torch.zeros(max(int(nbytes), 256), device=device, dtype=torch.uint8)
        _own_ws[key] = ws
    return ws


def _prep(*ts):
    out = []
    for t in ts:
        if t is None:
            out.append(None)
            continue
        if not t.is_cuda:
            raise native.NativeError("decode needs HIP tensors (got %s); there is no CPU path" % t.device)
        native.require_f32(t)
        out.append(t.contiguous())
    return out


def _expect(name, t, B, C, H, W, device):
    """The decoders read their maps through raw pointers: a map of the wrong size would be a
    silent out-of-bounds device read, where the reference's torch ops raise."""
    if t is None:
        return
    if t.dim() != 4 or tuple(t.shape) != (B, C, H, W):
        raise RuntimeError("%s must have shape %s, got %s" % (name, (B, C, H, W), tuple(t.shape)))
    if t.device != device:
        raise RuntimeError("%s is on %s, the heat-map on %s" % (name, t.device, device))


def ctdet_decode(heat, wh, reg=None, cat_spec_wh=False, K=100, apply_sigmoid=False,
                 return_inds=False, _debug_flags=0):
    heat, wh, reg = _prep(heat, wh, reg)
    lib = native.lib()
    if heat.dim() != 4:
        raise RuntimeError("heat must be (B, C, H, W)")
    B, C, H, W = heat.shape
    _expect("wh", wh, B, 2 * C if cat_spec_wh else 2, H, W, heat.device)
    _expect("reg", reg, B, 2, H, W, heat.device)
    if K > H * W:
        raise RuntimeError("selected index k out of range")  # torch.topk's error
    dets = torch.empty((B, K, 6), device=heat.device, dtype=torch.float32)
    inds = torch.empty((B, K), device=heat.device, dtype=torch.int32)
    nbytes = lib.cn_ctdet_decode_workspace_bytes(B, C, H, W, K)
    flags = int(bool(apply_sigmoid)) | int(_debug_flags)
    if _debug_flags:
        ws = _workspace(nbytes, heat.device)
    else:
        ws = _own_workspace("ctdet", nbytes, heat.device, (B, C, H, W, K))
        flags |= _STATE_CLEAN
    rc = lib.cn_ctdet_decode_f32(native.ptr(heat), native.ptr(wh), native.ptr(reg), B, C, H, W, K,
                                 int(bool(cat_spec_wh)), flags,
                                 native.ptr(dets), native.ptr(inds), native.ptr(ws), ws.numel(),
                                 native.stream_ptr())
    if rc:
        # a failed call may have left the image state words of an owned workspace dirty: never reuse it
        _own_ws.clear()
    native.check(rc, "cn_ctdet_decode_f32")
    if _verify_state and not _debug_flags:
        # debug mode (CN_DECODE_VERIFY_STATE=1): synchronise and look at the state words the kernel
        # promises to leave at zero (cn_decode_state_region)
        import ctypes
        off, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
        native.check(lib.cn_decode_state_region(B, C, H, W, K, ctypes.byref(off), ctypes.byref(nb)),
                     "cn_decode_state_region")
        words = ws[off.value:off.value + nb.value].view(torch.int32)
        if nb.value and int(words.abs().max().item()) != 0:
            _own_ws.clear()
            raise native.NativeError("cn_ctdet_decode_f32 left its workspace state words dirty")
    return (dets, inds.long()) if return_inds else dets


_NO_PEAK_TEST = 512   # flag bit of the decode entry points: rank every cell (plain topk)
_EXCT_CLAMP_ONE = 2   # CN_EXCT_CLAMP_ONE (cn_exct_decode_f32): clamp the peak-tested edge maps to 1


def _topk_channel(scores, K=40, apply_sigmoid=False, nms=False):
    """decode.py:92-101: per-channel top-K -> (scores, inds, ys, xs), each (B, C, K).  With
    ``nms=True`` the 3x3 peak test of ``_nms`` is fused in front (how the reference chains
    them, decode.py:528-533); ``nms=False`` is the plain function on any float map."""
    (scores,) = _prep(scores)
    lib = native.lib()
    B, C, H, W = scores.shape
    if K > H * W:
        raise RuntimeError("selected index k out of range")
    flags = int(bool(apply_sigmoid)) | (0 if nms else _NO_PEAK_TEST)
    s = torch.empty((B, C, K), device=scores.device, dtype=torch.float32)
    i = torch.empty((B, C, K), device=scores.device, dtype=torch.int32)
    nbytes = lib.cn_ctdet_decode_workspace_bytes(B, C, H, W, K)
    ws = _workspace(nbytes, scores.device)
    rc = lib.cn_nms_topk_channel_f32(native.ptr(scores), B, C, H, W, K, flags,
                                     native.ptr(s), native.ptr(i), native.ptr(ws), ws.numel(),
                                     native.stream_ptr())
    native.check(rc, "cn_nms_topk_channel_f32")
    i = i.long()
    ys = (i // W).float()
    xs = (i % W).float()
    return s, i, ys, xs


def multi_pose_decode(heat, wh, kps, reg=None, hm_hp=None, hp_offset=None, K=100,
                      apply_sigmoid=False):
    heat, wh, kps, reg, hm_hp, hp_offset = _prep(heat, wh, kps, reg, hm_hp, hp_offset)
    lib = native.lib()
    if heat.dim() != 4 or kps.dim() != 4 or kps.shape[1] % 2:
        raise RuntimeError("heat must be (B, C, H, W) and kps (B, 2J, H, W)")
    B, C, H, W = heat.shape
    J = kps.shape[1] // 2
    _expect("wh", wh, B, 2, H, W, heat.device)
    _expect("kps", kps, B, 2 * J, H, W, heat.device)
    _expect("reg", reg, B, 2, H, W, heat.device)
    _expect("hm_hp", hm_hp, B, J, H, W, heat.device)
    _expect("hp_offset", hp_offset, B, 2, H, W, heat.device)
    if K > H * W:
        raise RuntimeError("selected index k out of range")
    dets = torch.empty((B, K, 4 + 1 + 2 * J + 1), device=heat.device, dtype=torch.float32)
    nbytes = lib.cn_multi_pose_decode_workspace_bytes(B, C, H, W, J, K)
    ws = _workspace(nbytes, heat.device)
    rc = lib.cn_multi_pose_decode_f32(native.ptr(heat), native.ptr(wh), native.ptr(kps),
                                      native.ptr(reg), native.ptr(hm_hp), native.ptr(hp_offset),
                                      B, C, H, W, J, K, int(bool(apply_sigmoid)), native.ptr(dets),
                                      native.ptr(ws), ws.numel(), native.stream_ptr())
    native.check(rc, "cn_multi_pose_decode_f32")
    return dets


def _topk(scores, K=40, apply_sigmoid=False, nms=False):
    """decode.py:103-119: (scores, inds, clses, ys, xs), each (B, K); ``nms=True`` fuses the
    3x3 peak test of ``_nms`` in front (how the reference chains them), ``nms=False`` is the
    plain function."""
    (scores,) = _prep(scores)
    lib = native.lib()
    B, C, H, W = scores.shape
    if K > H * W:
        raise RuntimeError("selected index k out of range")
    flags = int(bool(apply_sigmoid)) | (0 if nms else _NO_PEAK_TEST)
    s = torch.empty((B, K), device=scores.device, dtype=torch.float32)
    i = torch.empty((B, K), device=scores.device, dtype=torch.int32)
    c = torch.empty((B, K), device=scores.device, dtype=torch.int32)
    ws = _workspace(lib.cn_ctdet_decode_workspace_bytes(B, C, H, W, K), scores.device)
    rc = lib.cn_topk_f32(native.ptr(scores), B, C, H, W, K, flags, native.ptr(s),
                         native.ptr(i), native.ptr(c), native.ptr(ws), ws.numel(),
                         native.stream_ptr())
    native.check(rc, "cn_topk_f32")
    i = i.long()
    return s, i, c.int(), (i // W).float(), (i % W).float()


def _transpose_and_gather_feat(feat, ind):
    """models/utils.py:21-26: (B,C,H,W), (B,K) -> (B,K,C), without the full-tensor transpose."""
    (feat,) = _prep(feat)
    lib = native.lib()
    B, C, H, W = feat.shape
    if ind.dim() != 2 or ind.shape[0] != B or ind.device != feat.device:
        raise RuntimeError("ind must be (B, K) on the device of feat")
    K = ind.shape[1]
    if ind.numel() and (int(ind.min()) < 0 or int(ind.max()) >= H * W):   # torch.gather raises
        raise RuntimeError("index out of range for a %dx%d map" % (H, W))
    ind32 = ind.to(torch.int32).contiguous()
    out = torch.empty((B, K, C), device=feat.device, dtype=torch.float32)
    rc = lib.cn_gather_feat_f32(native.ptr(feat), native.ptr(ind32), native.ptr(out), B, C, H, W, K,
                                native.stream_ptr())
    native.check(rc, "cn_gather_feat_f32")
    return out


def ddd_decode(heat, rot, depth, dim, wh=None, reg=None, K=40, apply_sigmoid=False):
    """decode.py:426-462 -> (B, K, 16) or (B, K, 18) with wh."""
    heat, rot, depth, dim, wh, reg = _prep(heat, rot, depth, dim, wh, reg)
    lib = native.lib()
    B, C, H, W = heat.shape
    if K > H * W:
        raise RuntimeError("selected index k out of range")
    _expect("rot", rot, B, 8, H, W, heat.device)
    _expect("depth", depth, B, 1, H, W, heat.device)
    _expect("dim", dim, B, 3, H, W, heat.device)
    _expect("wh", wh, B, 2, H, W, heat.device)
    _expect("reg", reg, B, 2, H, W, heat.device)
    dets = torch.empty((B, K, 18 if wh is not None else 16), device=heat.device, dtype=torch.float32)
    ws = _workspace(lib.cn_ddd_decode_workspace_bytes(B, C, H, W, K), heat.device)
    rc = lib.cn_ddd_decode_f32(native.ptr(heat), native.ptr(rot), native.ptr(depth), native.ptr(dim),
                               native.ptr(wh), native.ptr(reg), B, C, H, W, K,
                               int(bool(apply_sigmoid)), native.ptr(dets), native.ptr(ws),
                               ws.numel(), native.stream_ptr())
    native.check(rc, "cn_ddd_decode_f32")
    return dets


def agnex_ct_decode(t_heat, l_heat, b_heat, r_heat, ct_heat, t_regr=None, l_regr=None, b_regr=None,
                    r_regr=None, K=40, scores_thresh=0.1, center_thresh=0.1, aggr_weight=0.0,
                    num_dets=1000):
    """decode.py:121-271 -> (B, num_dets, 14): the class-agnostic ``exct_decode`` -- single-channel edge
    maps, the centre map over the classes; groupings are scored against the centre map's per-cell maximum
    and take its arg-max at the box centre as their class (``cn_agnex_ct_decode_f32``)."""
    return exct_decode(t_heat, l_heat, b_heat, r_heat, ct_heat, t_regr, l_regr, b_regr, r_regr, K=K,
                       scores_thresh=scores_thresh, center_thresh=center_thresh, aggr_weight=aggr_weight,
                       num_dets=num_dets, _agnostic=True)


def exct_decode(t_heat, l_heat, b_heat, r_heat, ct_heat, t_regr=None, l_regr=None, b_regr=None,
                r_regr=None, K=40, scores_thresh=0.1, center_thresh=0.1, aggr_weight=0.0,
                num_dets=1000, _agnostic=False):
    """decode.py:273-424 -> (B, num_dets, 14).  Heat-maps post-sigmoid (values <= 1).
    ``aggr_weight > 0``: the edge aggregation of decode.py:17-90,136-140 runs in front
    (``cn_exct_aggregate_f32``: rows of t / b, columns of l / r)."""
    t_heat, l_heat, b_heat, r_heat, ct_heat, t_regr, l_regr, b_regr, r_regr = _prep(
        t_heat, l_heat, b_heat, r_heat, ct_heat, t_regr, l_regr, b_regr, r_regr)
    lib = native.lib()
    B, C, H, W = t_heat.shape
    if _agnostic and C != 1:
        raise RuntimeError("agnex_ct_decode takes single-channel edge maps, got %d channels" % C)
    if aggr_weight > 0:
        for name, t in (("l_heat", l_heat), ("b_heat", b_heat), ("r_heat", r_heat)):
            _expect(name, t, B, C, H, W, t_heat.device)
        agg = []
        for t, horizontal in ((t_heat, 1), (l_heat, 0), (b_heat, 1), (r_heat, 0)):
            o = torch.empty_like(t)
            native.check(lib.cn_exct_aggregate_f32(native.ptr(t), native.ptr(o), B, C, H, W, horizontal,
                                                   float(aggr_weight), native.stream_ptr()),
                         "cn_exct_aggregate_f32")
            agg.append(o)
        t_heat, l_heat, b_heat, r_heat = agg
    for name, t in (("l_heat", l_heat), ("b_heat", b_heat), ("r_heat", r_heat)):
        _expect(name, t, B, C, H, W, t_heat.device)
    if _agnostic:
        if ct_heat.dim() != 4 or ct_heat.shape[0] != B or tuple(ct_heat.shape[2:]) != (H, W) or \
                ct_heat.device != t_heat.device:
            raise RuntimeError("ct_heat must be (B, classes, H, W) on the device of the edge maps")
    else:
        _expect("ct_heat", ct_heat, B, C, H, W, t_heat.device)
    for name, t in (("t_regr", t_regr), ("l_regr", l_regr), ("b_regr", b_regr), ("r_regr", r_regr)):
        _expect(name, t, B, 2, H, W, t_heat.device)
    if K > H * W or num_dets > K ** 4:
        raise RuntimeError("selected index k out of range")
    dets = torch.empty((B, num_dets, 14), device=t_heat.device, dtype=torch.float32)
    if _agnostic:
        Cc = int(ct_heat.shape[1])
        ws = _workspace(lib.cn_agnex_ct_decode_workspace_bytes(B, Cc, H, W, K), t_heat.device)
        rc = lib.cn_agnex_ct_decode_f32(native.ptr(t_heat), native.ptr(l_heat), native.ptr(b_heat),
                                        native.ptr(r_heat), native.ptr(ct_heat), native.ptr(t_regr),
                                        native.ptr(l_regr), native.ptr(b_regr), native.ptr(r_regr), B, Cc, H,
                                        W, K, scores_thresh, center_thresh, num_dets,
                                        _EXCT_CLAMP_ONE if aggr_weight > 0 else 0, native.ptr(dets),
                                        native.ptr(ws), ws.numel(), native.stream_ptr())
        native.check(rc, "cn_agnex_ct_decode_f32")
        return dets
    ws = _workspace(lib.cn_exct_decode_workspace_bytes(B, C, H, W, K), t_heat.device)
    rc = lib.cn_exct_decode_f32(native.ptr(t_heat), native.ptr(l_heat), native.ptr(b_heat),
                                native.ptr(r_heat), native.ptr(ct_heat), native.ptr(t_regr),
                                native.ptr(l_regr), native.ptr(b_regr), native.ptr(r_regr), B, C, H,
                                W, K, scores_thresh, center_thresh, num_dets,
                                # aggregated edge maps exceed 1: clamp behind the peak test (decode.py:302-305)
                                _EXCT_CLAMP_ONE if aggr_weight > 0 else 0, native.ptr(dets),
                                native.ptr(ws), ws.numel(), native.stream_ptr())
    native.check(rc, "cn_exct_decode_f32")
    return dets
