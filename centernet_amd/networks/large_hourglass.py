"""Hourglass-104 (arch 'hourglass'): two stacked 5-level hourglasses, no DCN.

Reference: src/lib/models/networks/large_hourglass.py -- convolution (:17-30), residual
(:48-74), kp_module (:117-174), exkp (:176-274), HourglassNet (:283-296).  State-dict
names follow that file (pre.0.conv.weight, kps.1.low2.low2.up1.0.bn1.running_var,
cnvs_.0.1.weight, hm.1.0.conv.bias, hm.1.1.weight, ...).
"""
import torch
import torch.nn as nn

from ..engine import PlannedModule


class convolution(nn.Module):
    def __init__(self, k, inp_dim, out_dim, stride=1, with_bn=True):
        super().__init__()
        pad = (k - 1) // 2
        self.conv = nn.Conv2d(inp_dim, out_dim, (k, k), padding=(pad, pad),
                              stride=(stride, stride), bias=not with_bn)
        self.bn = nn.BatchNorm2d(out_dim) if with_bn else nn.Sequential()
        self.with_bn = with_bn

    def describe(self, pb, x):
        c = self.conv
        return pb.conv(x, c.weight, bias=c.bias, bn=self.bn if self.with_bn else None, relu=True,
                       stride=c.stride[0], padding=c.padding[0])


class residual(nn.Module):
    def __init__(self, k, inp_dim, out_dim, stride=1, with_bn=True):
        super().__init__()
        self.conv1 = nn.Conv2d(inp_dim, out_dim, (3, 3), padding=(1, 1), stride=(stride, stride),
                               bias=False)
        self.bn1 = nn.BatchNorm2d(out_dim)
        self.conv2 = nn.Conv2d(out_dim, out_dim, (3, 3), padding=(1, 1), bias=False)
        self.bn2 = nn.BatchNorm2d(out_dim)
        self.skip = nn.Sequential(
            nn.Conv2d(inp_dim, out_dim, (1, 1), stride=(stride, stride), bias=False),
            nn.BatchNorm2d(out_dim)) if stride != 1 or inp_dim != out_dim else nn.Sequential()
        self.stride = stride

    def describe(self, pb, x):
        # large_hourglass.py:65-74: relu(bn2(conv2(relu(bn1(conv1 x)))) + skip(x))
        skip = x
        if len(self.skip) > 0:
            skip = pb.conv(x, self.skip[0].weight, bn=self.skip[1], stride=self.stride)
        out = pb.conv(x, self.conv1.weight, bn=self.bn1, relu=True, stride=self.stride, padding=1)
        return pb.conv(out, self.conv2.weight, bn=self.bn2, relu=True, residual=skip, padding=1)


def _seq(pb, seq, x):
    for m in seq:
        x = m.describe(pb, x)
    return x


def make_layer(k, inp_dim, out_dim, modules, layer=convolution, **kwargs):
    layers = [layer(k, inp_dim, out_dim, **kwargs)]
    layers += [layer(k, out_dim, out_dim, **kwargs) for _ in range(1, modules)]
    return nn.Sequential(*layers)


def make_layer_revr(k, inp_dim, out_dim, modules, layer=convolution, **kwargs):
    layers = [layer(k, inp_dim, inp_dim, **kwargs) for _ in range(modules - 1)]
    layers.append(layer(k, inp_dim, out_dim, **kwargs))
    return nn.Sequential(*layers)


def make_hg_layer(kernel, dim0, dim1, mod, layer=convolution, **kwargs):
    # large_hourglass.py:277-280: the stride-2 residual replaces max-pooling
    layers = [layer(kernel, dim0, dim1, stride=2)]
    layers += [layer(kernel, dim1, dim1) for _ in range(mod - 1)]
    return nn.Sequential(*layers)


class kp_module(nn.Module):
    def __init__(self, n, dims, modules, layer=residual):
        super().__init__()
        self.n = n
        curr_mod, next_mod = modules[0], modules[1]
        curr_dim, next_dim = dims[0], dims[1]
        self.up1 = make_layer(3, curr_dim, curr_dim, curr_mod, layer=layer)
        self.max1 = nn.Sequential()
        self.low1 = make_hg_layer(3, curr_dim, next_dim, curr_mod, layer=layer)
        self.low2 = kp_module(n - 1, dims[1:], modules[1:], layer=layer) if n > 1 else \
            make_layer(3, next_dim, next_dim, next_mod, layer=layer)
        self.low3 = make_layer_revr(3, next_dim, curr_dim, curr_mod, layer=layer)
        self.up2 = nn.Upsample(scale_factor=2)

    def describe(self, pb, x):
        # large_hourglass.py:163-174; up2 (nearest x2) and the merge add are one kernel
        up1 = _seq(pb, self.up1, x)
        low1 = _seq(pb, self.low1, x)
        low2 = self.low2.describe(pb, low1) if isinstance(self.low2, kp_module) \
            else _seq(pb, self.low2, low1)
        low3 = _seq(pb, self.low3, low2)
        return pb.upsample2x_add(low3, add=up1)


class exkp(PlannedModule):
    def __init__(self, n, nstack, dims, modules, heads, cnv_dim=256):
        super().__init__()
        self.nstack = nstack
        self.heads = heads
        curr_dim = dims[0]
        self.pre = nn.Sequential(convolution(7, 3, 128, stride=2), residual(3, 128, 256, stride=2))
        self.kps = nn.ModuleList([kp_module(n, dims, modules, layer=residual)
                                  for _ in range(nstack)])
        self.cnvs = nn.ModuleList([convolution(3, curr_dim, cnv_dim) for _ in range(nstack)])
        self.inters = nn.ModuleList([residual(3, curr_dim, curr_dim) for _ in range(nstack - 1)])
        self.inters_ = nn.ModuleList([
            nn.Sequential(nn.Conv2d(curr_dim, curr_dim, (1, 1), bias=False),
                          nn.BatchNorm2d(curr_dim)) for _ in range(nstack - 1)])
        self.cnvs_ = nn.ModuleList([
            nn.Sequential(nn.Conv2d(cnv_dim, curr_dim, (1, 1), bias=False),
                          nn.BatchNorm2d(curr_dim)) for _ in range(nstack - 1)])
        for head in heads.keys():
            module = nn.ModuleList([
                nn.Sequential(convolution(3, cnv_dim, curr_dim, with_bn=False),
                              nn.Conv2d(curr_dim, heads[head], (1, 1))) for _ in range(nstack)])
            self.__setattr__(head, module)
            if 'hm' in head:
                for heat in module:
                    heat[-1].bias.data.fill_(-2.19)

    def describe(self, pb, image):
        # large_hourglass.py:250-274.  Only the LAST stack's heads are evaluated: the
        # detectors consume self.model(images)[-1] (detectors/ctdet.py:30) and the
        # intermediate-supervision heads of earlier stacks are training-only outputs.
        inter = _seq(pb, self.pre, image)
        cnv = None
        for ind in range(self.nstack):
            kp = self.kps[ind].describe(pb, inter)
            cnv = self.cnvs[ind].describe(pb, kp)
            if ind < self.nstack - 1:
                a = pb.conv(inter, self.inters_[ind][0].weight, bn=self.inters_[ind][1])
                inter = pb.conv(cnv, self.cnvs_[ind][0].weight, bn=self.cnvs_[ind][1],
                                residual=a, relu=True)
                inter = self.inters[ind].describe(pb, inter)
        last = self.nstack - 1
        pairs = {h: (getattr(self, h)[last][0].conv, getattr(self, h)[last][1]) for h in self.heads}
        return pb.heads_from_convs(cnv, pairs)


class HourglassNet(exkp):
    def __init__(self, heads, num_stacks=2):
        super().__init__(5, num_stacks, [256, 256, 384, 384, 384, 512], [2, 2, 2, 2, 2, 4], heads,
                         cnv_dim=256)


def get_large_hourglass_net(num_layers, heads, head_conv):
    """arch 'hourglass' (large_hourglass.py:298-300)."""
    return HourglassNet(heads, 2)
