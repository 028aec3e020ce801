"""ctypes binding of oracle/libcn_oracle.so (TEST INFRASTRUCTURE, see oracle/__init__.py).

numpy in, numpy out; every function mirrors one reference callable:

    dcn_v2_forward      <- DCNv2Function.forward   (DCNv2/dcn_v2_func.py:22-38)
    nms                 <- _nms                    (models/decode.py:9-15)
    topk / topk_channel <- _topk / _topk_channel   (models/decode.py:92-119)
    ctdet_decode        <- ctdet_decode            (models/decode.py:464-495)
    multi_pose_decode   <- multi_pose_decode       (models/decode.py:497-571)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcn_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("dcn_v2_oracle.c", "decode_oracle.c")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcn_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_dcn_v2_forward.restype = ctypes.c_int
        _lib.oracle_dcn_v2_im2col.restype = ctypes.c_int
        _lib.oracle_topk_channel.restype = ctypes.c_int
        _lib.oracle_topk.restype = ctypes.c_int
        _lib.oracle_ctdet_decode.restype = ctypes.c_int
        _lib.oracle_multi_pose_decode.restype = ctypes.c_int
        _lib.oracle_nms3x3.restype = None
        _lib.oracle_transpose_and_gather_feat.restype = None
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _optf(a):
    if a is None:
        return None, None
    return _f(a)


def out_hw(H, W, kh, kw, stride, pad, dil):
    """Output shape rule, DCNv2/src/dcn_v2_cuda.c:40-41."""
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    return Ho, Wo


def dcn_v2_forward(x, offset, mask, weight, bias, stride=1, pad=1, dil=1, dg=1,
                   acc_mode=0):
    x, xp = _f(x)
    offset, op = _f(offset)
    mask, mp = _f(mask)
    weight, wp = _f(weight)
    bias, bp = _f(bias)
    B, Cin, H, W = x.shape
    Cout, Cin_w, kh, kw = weight.shape
    if Cin_w != Cin:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (Cin, Cin_w))
    Ho, Wo = out_hw(H, W, kh, kw, stride, pad, dil)
    assert offset.shape == (B, dg * 2 * kh * kw, Ho, Wo), offset.shape
    assert mask.shape == (B, dg * kh * kw, Ho, Wo), mask.shape
    out = np.empty((B, Cout, Ho, Wo), dtype=np.float32)
    rc = lib().oracle_dcn_v2_forward(
        xp, wp, bp, op, mp, out.ctypes.data_as(_f32p),
        B, Cin, H, W, Cout, kh, kw, stride, stride, pad, pad, dil, dil, dg, acc_mode)
    if rc != 0:
        raise RuntimeError("oracle_dcn_v2_forward failed: %d" % rc)
    return out


def dcn_v2_im2col(x, offset, mask, kh=3, kw=3, stride=1, pad=1, dil=1, dg=1):
    """One sample: x (Cin,H,W) -> columns (Cin*kh*kw, Ho, Wo)."""
    x, xp = _f(x)
    offset, op = _f(offset)
    mask, mp = _f(mask)
    Cin, H, W = x.shape
    Ho, Wo = out_hw(H, W, kh, kw, stride, pad, dil)
    cols = np.empty((Cin * kh * kw, Ho, Wo), dtype=np.float32)
    rc = lib().oracle_dcn_v2_im2col(xp, op, mp, cols.ctypes.data_as(_f32p), Cin, H, W,
                                    kh, kw, stride, stride, pad, pad, dil, dil, dg)
    if rc != 0:
        raise RuntimeError("oracle_dcn_v2_im2col failed: %d" % rc)
    return cols


def nms(heat):
    heat, hp = _f(heat)
    B, C, H, W = heat.shape
    out = np.empty_like(heat)
    lib().oracle_nms3x3(hp, out.ctypes.data_as(_f32p), B * C, H, W)
    return out


def topk_channel(scores, K):
    scores, sp = _f(scores)
    B, C, H, W = scores.shape
    s = np.empty((B, C, K), np.float32)
    i = np.empty((B, C, K), np.int64)
    y = np.empty((B, C, K), np.float32)
    x = np.empty((B, C, K), np.float32)
    rc = lib().oracle_topk_channel(sp, B, C, H, W, K, s.ctypes.data_as(_f32p),
                                   i.ctypes.data_as(_i64p), y.ctypes.data_as(_f32p),
                                   x.ctypes.data_as(_f32p))
    if rc != 0:
        raise RuntimeError("selected index k out of range")
    return s, i, y, x


def topk(scores, K):
    scores, sp = _f(scores)
    B, C, H, W = scores.shape
    s = np.empty((B, K), np.float32)
    i = np.empty((B, K), np.int64)
    c = np.empty((B, K), np.int32)
    y = np.empty((B, K), np.float32)
    x = np.empty((B, K), np.float32)
    rc = lib().oracle_topk(sp, B, C, H, W, K, s.ctypes.data_as(_f32p),
                           i.ctypes.data_as(_i64p), c.ctypes.data_as(_i32p),
                           y.ctypes.data_as(_f32p), x.ctypes.data_as(_f32p))
    if rc != 0:
        raise RuntimeError("selected index k out of range")
    return s, i, c, y, x


def ctdet_decode(heat, wh, reg=None, cat_spec_wh=False, K=100, return_inds=False):
    """heat is POST-sigmoid, as in the reference (models/decode.py:467)."""
    heat, hp = _f(heat)
    wh, wp = _f(wh)
    reg, rp = _optf(reg)
    B, C, H, W = heat.shape
    dets = np.empty((B, K, 6), np.float32)
    inds = np.empty((B, K), np.int64)
    rc = lib().oracle_ctdet_decode(hp, wp, rp, B, C, H, W, K, int(bool(cat_spec_wh)),
                                   dets.ctypes.data_as(_f32p), inds.ctypes.data_as(_i64p))
    if rc != 0:
        raise RuntimeError("selected index k out of range")
    return (dets, inds) if return_inds else dets


def multi_pose_decode(heat, wh, kps, reg=None, hm_hp=None, hp_offset=None, K=100):
    heat, hp = _f(heat)
    wh, wp = _f(wh)
    kps, kp = _f(kps)
    reg, rp = _optf(reg)
    hm_hp, hhp = _optf(hm_hp)
    hp_offset, hop = _optf(hp_offset)
    B, C, H, W = heat.shape
    J = kps.shape[1] // 2
    dets = np.empty((B, K, 4 + 1 + 2 * J + 1), np.float32)
    rc = lib().oracle_multi_pose_decode(hp, wp, kp, rp, hhp, hop, B, C, H, W, J, K,
                                        dets.ctypes.data_as(_f32p))
    if rc != 0:
        raise RuntimeError("selected index k out of range")
    return dets


def transpose_and_gather_feat(feat, ind):
    """models/utils.py:21-26: (B,C,H,W), (B,K) int64 -> (B,K,C)."""
    feat = np.ascontiguousarray(feat, np.float32)
    B, C, H, W = feat.shape
    flat = feat.reshape(B, C, H * W)
    out = np.empty((B, ind.shape[1], C), np.float32)
    for b in range(B):
        out[b] = flat[b][:, ind[b]].T
    return out


def ddd_decode(heat, rot, depth, dim, wh=None, reg=None, K=40):
    """models/decode.py:426-462 (heat is post-sigmoid): _nms, _topk, centre + reg (or + 0.5),
    gather rot / depth / dim (/ wh), concatenate [xs, ys, score, rot, depth, dim, (wh,) cls]."""
    s, i, c, y, x = topk(nms(heat), K)
    B = heat.shape[0]
    if reg is not None:
        r = transpose_and_gather_feat(reg, i)
        xs = x.reshape(B, K, 1) + r[:, :, 0:1]
        ys = y.reshape(B, K, 1) + r[:, :, 1:2]
    else:
        xs = x.reshape(B, K, 1) + np.float32(0.5)
        ys = y.reshape(B, K, 1) + np.float32(0.5)
    parts = [xs, ys, s.reshape(B, K, 1), transpose_and_gather_feat(rot, i),
             transpose_and_gather_feat(depth, i), transpose_and_gather_feat(dim, i)]
    if wh is not None:
        parts.append(transpose_and_gather_feat(wh, i))
    parts.append(c.reshape(B, K, 1).astype(np.float32))
    return np.concatenate(parts, axis=2).astype(np.float32)


def _edge_aggregate(heat, axis, weight):
    """models/decode.py:17-77: along ``axis`` (3 = x: _h_aggregate, 2 = y: _v_aggregate), from both
    ends, a running sum that continues while the values do not fall -- ret[i] = heat[i] + ret[i-1] *
    (heat[i] >= heat[i-1]) -- minus the value itself; result weight*first + weight*second + heat,
    float32, in the reference's order of operations."""
    f32 = np.float32
    h = np.moveaxis(np.ascontiguousarray(heat, f32), axis, 0)      # (N, ...)
    n = h.shape[0]
    fwd = h.copy()
    for i in range(1, n):                                          # _left / _top
        fwd[i] = fwd[i] + fwd[i - 1] * (h[i] >= h[i - 1]).astype(f32)
    bwd = h.copy()
    for i in range(n - 2, -1, -1):                                 # _right / _bottom
        bwd[i] = bwd[i] + bwd[i + 1] * (h[i] >= h[i + 1]).astype(f32)
    w = f32(weight)
    out = (w * (fwd - h) + w * (bwd - h)) + h
    return np.ascontiguousarray(np.moveaxis(out, 0, axis), f32)


def h_aggregate(heat, aggr_weight=0.1):
    """_h_aggregate (models/decode.py:71-73)."""
    return _edge_aggregate(heat, 3, aggr_weight)


def v_aggregate(heat, aggr_weight=0.1):
    """_v_aggregate (models/decode.py:75-77)."""
    return _edge_aggregate(heat, 2, aggr_weight)


def agnex_ct_decode(t_heat, l_heat, b_heat, r_heat, ct_heat, t_regr=None, l_regr=None, b_regr=None,
                    r_regr=None, K=40, scores_thresh=0.1, center_thresh=0.1, num_dets=1000, aggr_weight=0.0):
    """models/decode.py:121-271, the class-agnostic form: the statements of exct_decode with (B, 1, H, W)
    edge maps, ``torch.max(ct_heat, dim=1)`` as the centre map (:164), no class rule (:202-215) and the
    arg-max class at the box centre (:175-177,262-263)."""
    return exct_decode(t_heat, l_heat, b_heat, r_heat, ct_heat, t_regr, l_regr, b_regr, r_regr, K=K,
                       scores_thresh=scores_thresh, center_thresh=center_thresh, num_dets=num_dets,
                       aggr_weight=aggr_weight, agnostic=True)


def exct_decode(t_heat, l_heat, b_heat, r_heat, ct_heat, t_regr=None, l_regr=None, b_regr=None,
                r_regr=None, K=40, scores_thresh=0.1, center_thresh=0.1, num_dets=1000, aggr_weight=0.0,
                agnostic=False):
    """models/decode.py:273-424, statement by statement in float32 numpy (aggr_weight > 0: the edge
    aggregation of :136-140 in front).  Ties of the final top-k are ordered by candidate index (torch
    leaves that unspecified)."""
    f32 = np.float32
    B, C, H, W = t_heat.shape
    if aggr_weight > 0:                                              # :136-140
        t_heat, b_heat = h_aggregate(t_heat, aggr_weight), h_aggregate(b_heat, aggr_weight)
        l_heat, r_heat = v_aggregate(l_heat, aggr_weight), v_aggregate(r_heat, aggr_weight)
    tops = [topk(np.minimum(nms(h), f32(1.0)), K) for h in (t_heat, l_heat, b_heat, r_heat)]  # :297-310

    def ax(v, e):  # (B,K) -> broadcast along axis e of (B,K,K,K,K)
        shape = [B, 1, 1, 1, 1]
        shape[1 + e] = K
        return v.reshape(shape)
    S = [ax(tops[e][0], e) for e in range(4)]
    CL = [ax(tops[e][2], e) for e in range(4)]
    Y = [ax(tops[e][3], e) for e in range(4)]
    X = [ax(tops[e][4], e) for e in range(4)]
    t_ys, l_ys, b_ys, r_ys = Y
    t_xs, l_xs, b_xs, r_xs = X
    full = (B, K, K, K, K)
    box_ct_xs = (((l_xs + r_xs) + f32(0.5)) / f32(2)).astype(np.int64)          # :331
    box_ct_ys = (((t_ys + b_ys) + f32(0.5)) / f32(2)).astype(np.int64)          # :332
    if agnostic:                                                                 # :164,173-177
        ct_all = np.ascontiguousarray(ct_heat, f32)
        ct_clses = np.argmax(ct_all, axis=1).reshape(B, -1)                      # first maximum, as torch.max
        ct_flat = np.max(ct_all, axis=1).reshape(B, -1)
        ct_inds = np.broadcast_to(box_ct_ys * W + box_ct_xs, full).reshape(B, -1)
        agn_clses = np.take_along_axis(ct_clses, ct_inds, axis=1).reshape(full)
    else:
        ct_inds = CL[0].astype(np.int64) * (H * W) + box_ct_ys * W + box_ct_xs  # :333
        ct_inds = np.broadcast_to(ct_inds, full).reshape(B, -1)
        ct_flat = np.ascontiguousarray(ct_heat, f32).reshape(B, -1)
    ct_scores = np.take_along_axis(ct_flat, ct_inds, axis=1).reshape(full)      # :334-336
    scores = ((((S[0] + S[1]) + S[2]) + S[3]) + f32(2) * ct_scores) / f32(6)    # :343
    cls_inds = (CL[0] != CL[1]) | (CL[0] != CL[2]) | (CL[0] != CL[3])           # :346-348
    top_inds = (t_ys > l_ys) | (t_ys > b_ys) | (t_ys > r_ys)                     # :350-357
    left_inds = (l_xs > t_xs) | (l_xs > b_xs) | (l_xs > r_xs)
    bottom_inds = (b_ys < t_ys) | (b_ys < l_ys) | (b_ys < r_ys)
    right_inds = (r_xs < t_xs) | (r_xs < l_xs) | (r_xs < b_xs)
    th, cth = f32(scores_thresh), f32(center_thresh)
    sc_inds = (S[0] < th) | (S[1] < th) | (S[2] < th) | (S[3] < th) | (ct_scores < cth)  # :359-362
    scores = scores.astype(f32)
    rules = (sc_inds, top_inds, left_inds, bottom_inds, right_inds) if agnostic else \
        (sc_inds, cls_inds, top_inds, left_inds, bottom_inds, right_inds)
    for flag in rules:                                                                    # :364-369 (:217-221)
        scores = scores - np.broadcast_to(flag, full).astype(f32)
    scores = scores.reshape(B, -1)
    order = np.lexsort((np.broadcast_to(np.arange(scores.shape[1]), scores.shape), -scores), axis=1)
    inds = order[:, :num_dets]                                                   # :372
    top_scores = np.take_along_axis(scores, inds, axis=1)[..., None]
    regs = (t_regr, l_regr, b_regr, r_regr)
    XS, YS = [], []
    for e in range(4):
        if all(r is not None for r in regs):                                     # :375-392
            g = transpose_and_gather_feat(regs[e], tops[e][1])
            XS.append(X[e] + ax(g[..., 0], e))
            YS.append(Y[e] + ax(g[..., 1], e))
        else:                                                                    # :393-401
            XS.append(X[e] + f32(0.5))
            YS.append(Y[e] + f32(0.5))

    def pick(v):
        return np.take_along_axis(np.broadcast_to(v, full).reshape(B, -1), inds, axis=1)[..., None]
    cols = [pick(XS[1]), pick(YS[0]), pick(XS[3]), pick(YS[2]), top_scores]       # bboxes :403
    for e in range(4):
        cols += [pick(XS[e]), pick(YS[e])]
    cols.append(pick(agn_clses if agnostic else CL[0]).astype(f32))
    return np.concatenate(cols, axis=2).astype(f32)
