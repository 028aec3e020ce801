"""oracle/net_oracle.py -- TEST INFRASTRUCTURE (see oracle/__init__.py).

torch-CPU functional restatement of the reference network graphs, driven by a
state-dict (so it shares no code with centernet_amd/networks).  Dense ops are
PyTorch's own (the reference's third-party arithmetic); DCN layers go through
oracle/dcn_v2_oracle.c.

Follows (relative to /root/reference/src/lib/models/networks):
  res_N     msra_resnet.py:107-226  (forward :213-226)
  resdcn_N  resnet_dcn.py:130-263   (forward :248-263, DCN.forward DCNv2/dcn_v2.py:64-70)
Pinned by tests/golden/net_*.npz, produced by running the reference's own module
classes on CPU (tests/golden/gen_golden.py; DCN.forward there is routed to the C
oracle because the reference's extension cannot be built).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import cref

EPS = 1e-5


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.0, EPS)


def dcn(x, sd, p, stride=1, padding=1, dilation=1, deformable_groups=1):
    """DCN.forward, DCNv2/dcn_v2.py:64-70 (conv_offset_mask has the DCN's stride and padding,
    :52-57)."""
    out = F.conv2d(x, sd[p + ".conv_offset_mask.weight"], sd[p + ".conv_offset_mask.bias"],
                   stride=stride, padding=padding)
    o1, o2, mask = torch.chunk(out, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    mask = torch.sigmoid(mask)
    y = cref.dcn_v2_forward(x.numpy(), offset.numpy(), mask.numpy(), sd[p + ".weight"].numpy(),
                            sd[p + ".bias"].numpy(), stride, padding, dilation, deformable_groups)
    return torch.from_numpy(y)


TRACE = None  # set to a list to record intermediate activations (debugging aid)


def _t(name, x):
    if TRACE is not None:
        TRACE.append((name, x))
    return x


def _basic_block(x, sd, p, stride):
    res = x
    if (p + ".downsample.0.weight") in sd:
        res = _t(p + ".ds", _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0), sd,
                                p + ".downsample.1"))
    out = _t(p + ".c1", F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), sd, p + ".bn1")))
    out = _bn(F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".bn2")
    return _t(p + ".out", F.relu(out + res))


def _bottleneck(x, sd, p, stride):
    res = x
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
    out = F.relu(_bn(F.conv2d(out, sd[p + ".conv2.weight"], None, stride, 1), sd, p + ".bn2"))
    out = _bn(F.conv2d(out, sd[p + ".conv3.weight"]), sd, p + ".bn3")
    if (p + ".downsample.0.weight") in sd:
        res = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0), sd,
                  p + ".downsample.1")
    return F.relu(out + res)


_SPEC = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottle", [3, 4, 6, 3]),
         101: ("bottle", [3, 4, 23, 3]), 152: ("bottle", [3, 8, 36, 3])}


def _head(x, sd, h):
    if (h + ".0.weight") in sd:
        k = sd[h + ".0.weight"].shape[-1]
        y = F.relu(F.conv2d(x, sd[h + ".0.weight"], sd[h + ".0.bias"], 1, k // 2))
        last = max(int(key.split(".")[1]) for key in sd if key.startswith(h + ".") and
                   key.split(".")[1].isdigit())
        k2 = sd["%s.%d.weight" % (h, last)].shape[-1]
        return F.conv2d(y, sd["%s.%d.weight" % (h, last)], sd["%s.%d.bias" % (h, last)], 1, k2 // 2)
    k = sd[h + ".weight"].shape[-1]
    return F.conv2d(x, sd[h + ".weight"], sd[h + ".bias"], 1, k // 2)


def resnet_forward(sd, x, num_layers, heads, dcn_up):
    """``heads``: iterable of head names.  Returns {head: (B,C,H/4,W/4) tensor}."""
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    kind, layers = _SPEC[num_layers]
    block = _basic_block if kind == "basic" else _bottleneck
    with torch.no_grad():
        x = _t("stem", F.relu(_bn(F.conv2d(x, sd["conv1.weight"], None, 2, 3), sd, "bn1")))
        x = _t("pool", F.max_pool2d(x, 3, 2, 1))
        for li, n in enumerate(layers):
            for bi in range(n):
                stride = 2 if (li > 0 and bi == 0) else 1
                x = block(x, sd, "layer%d.%d" % (li + 1, bi), stride)
        if dcn_up:  # resnet_dcn.py:209-246: [DCN, BN, ReLU, ConvT, BN, ReLU] x 3
            for i in range(3):
                b = 6 * i
                x = _t("dcn%d" % i, F.relu(_bn(dcn(x, sd, "deconv_layers.%d" % b), sd,
                                               "deconv_layers.%d" % (b + 1))))
                x = F.conv_transpose2d(x, sd["deconv_layers.%d.weight" % (b + 3)], None, 2, 1, 0)
                x = _t("up%d" % i, F.relu(_bn(x, sd, "deconv_layers.%d" % (b + 4))))
        else:  # msra_resnet.py: [ConvT, BN, ReLU] x 3
            for i in range(3):
                b = 3 * i
                x = F.conv_transpose2d(x, sd["deconv_layers.%d.weight" % b], None, 2, 1, 0)
                x = _t("up%d" % i, F.relu(_bn(x, sd, "deconv_layers.%d" % (b + 1))))
        return {h: _head(x, sd, h) for h in heads}


# ----------------------------------------------------------------------------- DLA-34
# pose_dla_dcn.py: BasicBlock :31-62, Root :147-165, Tree :168-221, DLA :224-286,
# DeformConv :345-357, IDAUp :360-386, DLAUp :390-413, DLASeg.forward :470-482
_DLA_LEVELS = [1, 1, 1, 2, 2, 1]
_DLA_CH = [16, 32, 64, 128, 256, 512]


def _dla_block(x, sd, p, stride, residual=None):
    if residual is None:
        residual = x
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), sd, p + ".bn1"))
    out = _bn(F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".bn2")
    return F.relu(out + residual)


def _dla_tree(x, sd, p, levels, stride, level_root, residual=None, children=None):
    children = [] if children is None else children
    bottom = F.max_pool2d(x, stride, stride) if stride > 1 else x
    if (p + ".project.0.weight") in sd:
        residual = _t(p + ".project", _bn(F.conv2d(bottom, sd[p + ".project.0.weight"]), sd,
                                          p + ".project.1"))
    else:
        residual = bottom
    if level_root:
        children.append(bottom)
    if levels == 1:
        x1 = _dla_block(x, sd, p + ".tree1", stride, residual)
        x2 = _dla_block(x1, sd, p + ".tree2", 1)
        cat = torch.cat([x2, x1] + children, 1)
        y = _bn(F.conv2d(cat, sd[p + ".root.conv.weight"]), sd, p + ".root.bn")
        return _t(p + ".root", F.relu(y))  # root_residual is False for dla34 (:310-312)
    x1 = _dla_tree(x, sd, p + ".tree1", levels - 1, stride, False, residual)
    children.append(x1)
    return _dla_tree(x1, sd, p + ".tree2", levels - 1, 1, False, children=children)


def _deform_conv(x, sd, p):
    return F.relu(_bn(dcn(x, sd, p + ".conv"), sd, p + ".actf.0"))


def _ida_up(layers, sd, p, startp, endp):
    for i in range(startp + 1, endp):
        k = str(i - startp)
        w = sd[p + ".up_" + k + ".weight"]
        f = w.shape[2] // 2
        y = _deform_conv(layers[i], sd, p + ".proj_" + k)
        y = F.conv_transpose2d(y, w, None, stride=f, padding=f // 2, output_padding=0,
                               groups=w.shape[0])
        layers[i] = _t(p + ".node_" + k, _deform_conv(y + layers[i - 1], sd, p + ".node_" + k))


def dla34_forward(sd, x, heads, down_ratio=4, last_level=5):
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    first_level = {2: 1, 4: 2, 8: 3, 16: 4}[down_ratio]
    with torch.no_grad():
        x = F.relu(_bn(F.conv2d(x, sd["base.base_layer.0.weight"], None, 1, 3), sd,
                       "base.base_layer.1"))
        y = []
        for i in range(6):
            p = "base.level%d" % i
            if i < 2:
                x = F.relu(_bn(F.conv2d(x, sd[p + ".0.weight"], None, 2 if i == 1 else 1, 1), sd,
                               p + ".1"))
            else:
                x = _dla_tree(x, sd, p, _DLA_LEVELS[i], 2, i >= 3)
            y.append(_t("level%d" % i, x))
        # DLAUp.forward
        layers = list(y)
        out = [layers[-1]]
        for i in range(len(layers) - first_level - 1):
            _ida_up(layers, sd, "dla_up.ida_%d" % i, len(layers) - i - 2, len(layers))
            out.insert(0, layers[-1])
        z = [out[i].clone() for i in range(last_level - first_level)]
        _ida_up(z, sd, "ida_up", 0, len(z))
        return {h: _head(z[-1], sd, h) for h in heads}


# ----------------------------------------------------------------------------- Hourglass-104
# large_hourglass.py: convolution :17-30, residual :48-74, kp_module.forward :163-174,
# exkp.forward :250-274, HourglassNet :283-296
_HG_MODULES = [2, 2, 2, 2, 2, 4]


def _hg_conv(x, sd, p, k, stride=1, with_bn=True):
    y = F.conv2d(x, sd[p + ".conv.weight"], sd.get(p + ".conv.bias"), stride, (k - 1) // 2)
    if with_bn:
        y = _bn(y, sd, p + ".bn")
    return F.relu(y)


def _hg_res(x, sd, p, stride=1):
    y = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), sd, p + ".bn1"))
    y = _bn(F.conv2d(y, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".bn2")
    skip = x
    if (p + ".skip.0.weight") in sd:
        skip = _bn(F.conv2d(x, sd[p + ".skip.0.weight"], None, stride, 0), sd, p + ".skip.1")
    return F.relu(y + skip)


def _hg_seq(x, sd, p, n, first_stride=1):
    for i in range(n):
        x = _hg_res(x, sd, "%s.%d" % (p, i), first_stride if i == 0 else 1)
    return x


def _hg_kp(x, sd, p, n, mods):
    up1 = _hg_seq(x, sd, p + ".up1", mods[0])
    low1 = _hg_seq(x, sd, p + ".low1", mods[0], first_stride=2)   # max1 is an empty Sequential
    if n > 1:
        low2 = _hg_kp(low1, sd, p + ".low2", n - 1, mods[1:])
    else:
        low2 = _hg_seq(low1, sd, p + ".low2", mods[1])
    low3 = _hg_seq(low2, sd, p + ".low3", mods[0])
    up2 = F.interpolate(low3, scale_factor=2, mode="nearest")       # nn.Upsample(scale_factor=2)
    return _t(p, up1 + up2)


def hourglass_forward(sd, x, heads, nstack=2, all_stacks=False):
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    with torch.no_grad():
        inter = _hg_conv(x, sd, "pre.0", 7, 2)
        inter = _t("pre", _hg_res(inter, sd, "pre.1", 2))
        outs = []
        for ind in range(nstack):
            kp = _hg_kp(inter, sd, "kps.%d" % ind, 5, _HG_MODULES)
            cnv = _t("cnv%d" % ind, _hg_conv(kp, sd, "cnvs.%d" % ind, 3))
            if all_stacks or ind == nstack - 1:
                out = {}
                for h in heads:
                    y = _hg_conv(cnv, sd, "%s.%d.0" % (h, ind), 3, with_bn=False)
                    out[h] = F.conv2d(y, sd["%s.%d.1.weight" % (h, ind)], sd["%s.%d.1.bias" % (h, ind)])
                outs.append(out)
            if ind < nstack - 1:
                a = _bn(F.conv2d(inter, sd["inters_.%d.0.weight" % ind]), sd, "inters_.%d.1" % ind)
                b = _bn(F.conv2d(cnv, sd["cnvs_.%d.0.weight" % ind]), sd, "cnvs_.%d.1" % ind)
                inter = F.relu(a + b)
                inter = _t("inter%d" % ind, _hg_res(inter, sd, "inters.%d" % ind))
        return outs if all_stacks else outs[-1]


def forward(arch, sd, x, heads):
    """arch string as in the reference's create_model ('res_18', 'resdcn_18', ...)."""
    name, _, n = arch.partition("_")
    n = int(n) if n else 0
    if name == "res":
        return resnet_forward(sd, x, n, heads, dcn_up=False)
    if name == "resdcn":
        return resnet_forward(sd, x, n, heads, dcn_up=True)
    if name == "dla" and n == 34:
        return dla34_forward(sd, x, heads)
    if name == "hourglass":
        return hourglass_forward(sd, x, heads)
    raise NotImplementedError(arch)


def _flip_w(t):
    """flip_tensor, models/utils.py:28-29."""
    return torch.flip(t, [3])


def _flip_lr(x, flip_idx):
    """models/utils.py:33-39 (NumPy round trip, as the reference does it)."""
    tmp = x.numpy()[..., ::-1].copy()
    shape = tmp.shape
    for e in flip_idx:
        tmp[:, e[0], ...], tmp[:, e[1], ...] = tmp[:, e[1], ...].copy(), tmp[:, e[0], ...].copy()
    return torch.from_numpy(tmp.reshape(shape))


def _flip_lr_off(x, flip_idx):
    """models/utils.py:41-50."""
    tmp = x.numpy()[..., ::-1].copy()
    shape = tmp.shape
    tmp = tmp.reshape(tmp.shape[0], 17, 2, tmp.shape[2], tmp.shape[3])
    tmp[:, :, 0, :, :] *= -1
    for e in flip_idx:
        tmp[:, e[0], ...], tmp[:, e[1], ...] = tmp[:, e[1], ...].copy(), tmp[:, e[0], ...].copy()
    return torch.from_numpy(tmp.reshape(shape))


def ctdet_process(arch, sd, images, heads, K=100, reg_offset=True, cat_spec_wh=False,
                  flip_test=False):
    """CtdetDetector.process (detectors/ctdet.py:28-45): returns (output, dets).  With
    ``flip_test`` the batch is [frame, mirrored frame] (base_detector.py:59-60)."""
    out = forward(arch, sd, images, heads)
    hm = out["hm"].sigmoid_()
    wh = out["wh"]
    reg = out["reg"] if reg_offset and "reg" in out else None
    if flip_test:
        hm = (hm[0:1] + _flip_w(hm[1:2])) / 2
        wh = (wh[0:1] + _flip_w(wh[1:2])) / 2
        reg = reg[0:1] if reg is not None else None
    dets = cref.ctdet_decode(hm.numpy(), wh.numpy(), None if reg is None else reg.numpy(),
                             cat_spec_wh=cat_spec_wh, K=K)
    return out, dets


COCO_FLIP_IDX = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]


def multi_pose_process(arch, sd, images, heads, K=100, flip_test=False, flip_idx=None):
    """MultiPoseDetector.process (detectors/multi_pose.py:29-60) with the default options
    (reg_offset, hm_hp, reg_hp_offset on; mse_loss off): returns (output, dets)."""
    flip_idx = COCO_FLIP_IDX if flip_idx is None else flip_idx
    out = forward(arch, sd, images, heads)
    out["hm"] = out["hm"].sigmoid_()
    out["hm_hp"] = out["hm_hp"].sigmoid_()
    reg, hm_hp, hp_offset = out["reg"], out["hm_hp"], out["hp_offset"]
    if flip_test:
        out["hm"] = (out["hm"][0:1] + _flip_w(out["hm"][1:2])) / 2
        out["wh"] = (out["wh"][0:1] + _flip_w(out["wh"][1:2])) / 2
        out["hps"] = (out["hps"][0:1] + _flip_lr_off(out["hps"][1:2], flip_idx)) / 2
        hm_hp = (hm_hp[0:1] + _flip_lr(hm_hp[1:2], flip_idx)) / 2
        reg, hp_offset = reg[0:1], hp_offset[0:1]
    dets = cref.multi_pose_decode(out["hm"].numpy(), out["wh"].numpy(), out["hps"].numpy(),
                                  reg.numpy(), hm_hp.numpy(), hp_offset.numpy(), K=K)
    return out, dets


def ddd_process(arch, sd, images, heads, K=100, reg_bbox=True, reg_offset=True):
    """DddDetector.process (detectors/ddd.py:56-73): returns (output, dets (B, K, 16 | 18))."""
    out = forward(arch, sd, images, heads)
    out["hm"] = out["hm"].sigmoid_()
    out["dep"] = 1. / (out["dep"].sigmoid() + 1e-6) - 1.
    wh = out["wh"].numpy() if reg_bbox else None
    reg = out["reg"].numpy() if reg_offset else None
    dets = cref.ddd_decode(out["hm"].numpy(), out["rot"].numpy(), out["dep"].numpy(), out["dim"].numpy(),
                           wh=wh, reg=reg, K=K)
    return out, dets


def exdet_process(arch, sd, images, heads, K=40, scores_thresh=0.1, center_thresh=0.1, aggr_weight=0.0,
                  reg_offset=True, agnostic=False):
    """ExdetDetector.process (detectors/exdet.py:28-55): returns (output, dets (B, 1000, 14));
    ``agnostic``: agnex_ct_decode instead of exct_decode (exdet.py:26)."""
    out = forward(arch, sd, images, heads)
    heats = [out[n].sigmoid_().numpy() for n in ("hm_t", "hm_l", "hm_b", "hm_r", "hm_c")]
    regs = [out[n].numpy() for n in ("reg_t", "reg_l", "reg_b", "reg_r")] if reg_offset else [None] * 4
    decode = cref.agnex_ct_decode if agnostic else cref.exct_decode
    dets = decode(*(heats + regs), K=K, scores_thresh=scores_thresh, center_thresh=center_thresh,
                  aggr_weight=aggr_weight)
    return out, dets
