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


def dcn(x, sd, p):
    """DCN.forward, DCNv2/dcn_v2.py:64-70."""
    out = F.conv2d(x, sd[p + ".conv_offset_mask.weight"], sd[p + ".conv_offset_mask.bias"],
                   stride=1, padding=1)
    o1, o2, mask = torch.chunk(out, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    mask = torch.sigmoid(mask)
    y = cref.dcn_v2_forward(x.numpy(), offset.numpy(), mask.numpy(), sd[p + ".weight"].numpy(),
                            sd[p + ".bias"].numpy(), 1, 1, 1, 1)
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


def forward(arch, sd, x, heads):
    """arch string as in the reference's create_model ('res_18', 'resdcn_18', ...)."""
    name, _, n = arch.partition("_")
    n = int(n) if n else 0
    if name == "res":
        return resnet_forward(sd, x, n, heads, dcn_up=False)
    if name == "resdcn":
        return resnet_forward(sd, x, n, heads, dcn_up=True)
    raise NotImplementedError(arch)


def ctdet_process(arch, sd, images, heads, K=100, reg_offset=True, cat_spec_wh=False):
    """CtdetDetector.process without flip (detectors/ctdet.py:28-45): returns (output, dets)."""
    out = forward(arch, sd, images, heads)
    hm = out["hm"].sigmoid_()
    dets = cref.ctdet_decode(hm.numpy(), out["wh"].numpy(),
                             out["reg"].numpy() if reg_offset and "reg" in out else None,
                             cat_spec_wh=cat_spec_wh, K=K)
    return out, dets
