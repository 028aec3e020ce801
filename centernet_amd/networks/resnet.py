"""ResNet trunks with an up-sampling path and CenterNet heads.

Two reference files are covered by one class:
  arch 'res_N'    : src/lib/models/networks/msra_resnet.py  (ConvTranspose up-path)
  arch 'resdcn_N' : src/lib/models/networks/resnet_dcn.py   (DCN + ConvTranspose up-path)
Parameter names/shapes follow those files exactly (conv1, bn1, layer{1..4}.{i}.conv{1,2,3},
layer*.0.downsample.{0,1}, deconv_layers.{i}, <head>.{0,2}).
"""
import math

import torch
import torch.nn as nn

from ..dcn_v2 import DCN
from ..engine import PlannedModule

BN_MOMENTUM = 0.1


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride

    def describe(self, pb, x):
        # resnet_dcn.py:49-67
        res = x
        if self.downsample is not None:
            ds = self.downsample
            res = pb.conv(x, ds[0].weight, bn=ds[1], stride=ds[0].stride[0])
        out = pb.conv(x, self.conv1.weight, bn=self.bn1, relu=True, stride=self.stride, padding=1)
        return pb.conv(out, self.conv2.weight, bn=self.bn2, relu=True, residual=res, padding=1)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride

    def describe(self, pb, x):
        # resnet_dcn.py:88-108
        res = x
        if self.downsample is not None:
            ds = self.downsample
            res = pb.conv(x, ds[0].weight, bn=ds[1], stride=ds[0].stride[0])
        out = pb.conv(x, self.conv1.weight, bn=self.bn1, relu=True)
        out = pb.conv(out, self.conv2.weight, bn=self.bn2, relu=True, stride=self.stride, padding=1)
        return pb.conv(out, self.conv3.weight, bn=self.bn3, relu=True, residual=res)


def _bilinear_up_init(up):
    # resnet_dcn.py:110-119 (fill_up_weights): only w[:, 0] is filled
    w = up.weight.data
    f = math.ceil(w.size(2) / 2)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    for i in range(w.size(2)):
        for j in range(w.size(3)):
            w[0, 0, i, j] = (1 - math.fabs(i / f - c)) * (1 - math.fabs(j / f - c))
    for ch in range(1, w.size(0)):
        w[ch, 0, :, :] = w[0, 0, :, :]


class PoseResNet(PlannedModule):
    def __init__(self, block, layers, heads, head_conv, dcn=True):
        super().__init__()
        self.inplanes = 64
        self.heads = heads
        self.deconv_with_bias = False
        self.use_dcn = dcn
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        filters = [256, 128, 64] if dcn else [256, 256, 256]
        self.deconv_layers = self._make_deconv_layer(filters)
        feat = filters[-1]
        # msra_resnet.py:133 creates heads in sorted() order, resnet_dcn.py:155 in dict order;
        # only RNG-dependent init differs, the names are the same.
        for head in (self.heads if dcn else sorted(self.heads)):
            classes = self.heads[head]
            if head_conv > 0:
                fc = nn.Sequential(nn.Conv2d(feat, head_conv, 3, padding=1, bias=True),
                                   nn.ReLU(inplace=True),
                                   nn.Conv2d(head_conv, classes, 1, bias=True))
                last = fc[-1]
            else:
                fc = nn.Conv2d(feat, classes, 1, bias=True)
                last = fc
            if 'hm' in head:
                last.bias.data.fill_(-2.19)  # resnet_dcn.py:165-166
            elif dcn:
                for m in fc.modules():       # fill_fc_weights, resnet_dcn.py:121-128
                    if isinstance(m, nn.Conv2d):
                        nn.init.normal_(m.weight, std=0.001)
                        nn.init.constant_(m.bias, 0)
            self.__setattr__(head, fc)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion, momentum=BN_MOMENTUM))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _make_deconv_layer(self, filters):
        layers = []
        for planes in filters:
            if self.use_dcn:
                layers.append(DCN(self.inplanes, planes, kernel_size=(3, 3), stride=1, padding=1,
                                  dilation=1, deformable_groups=1))
                layers.append(nn.BatchNorm2d(planes, momentum=BN_MOMENTUM))
                layers.append(nn.ReLU(inplace=True))
                up = nn.ConvTranspose2d(planes, planes, 4, 2, 1, 0, bias=self.deconv_with_bias)
                _bilinear_up_init(up)
            else:
                up = nn.ConvTranspose2d(self.inplanes, planes, 4, 2, 1, 0,
                                        bias=self.deconv_with_bias)
            layers.append(up)
            layers.append(nn.BatchNorm2d(planes, momentum=BN_MOMENTUM))
            layers.append(nn.ReLU(inplace=True))
            self.inplanes = planes
        return nn.Sequential(*layers)

    def describe(self, pb, x):
        # resnet_dcn.py:248-263 / msra_resnet.py forward
        x = pb.conv(x, self.conv1.weight, bn=self.bn1, relu=True, stride=2, padding=3)
        x = pb.maxpool(x, 3, 2, 1)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                x = blk.describe(pb, x)
        mods = list(self.deconv_layers)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, DCN):
                x = pb.dcn(x, m, bn=mods[i + 1], relu=True)
                i += 3
            elif isinstance(m, nn.ConvTranspose2d):
                x = pb.conv_transpose4x4s2(x, m.weight, bn=mods[i + 1], relu=True)
                i += 3
            else:
                raise RuntimeError("unexpected module in deconv_layers: %r" % m)
        return pb.heads(x, {h: getattr(self, h) for h in self.heads})


resnet_spec = {18: (BasicBlock, [2, 2, 2, 2]),
               34: (BasicBlock, [3, 4, 6, 3]),
               50: (Bottleneck, [3, 4, 6, 3]),
               101: (Bottleneck, [3, 4, 23, 3]),
               152: (Bottleneck, [3, 8, 36, 3])}


def get_pose_net_dcn(num_layers, heads, head_conv=256):
    """arch 'resdcn_N' (resnet_dcn.py:285-290).  The reference downloads ImageNet weights
    here (model_zoo.load_url); this image has no network, so weights stay at their
    default initialisation until ``load_model`` supplies a checkpoint."""
    block, layers = resnet_spec[num_layers]
    return PoseResNet(block, layers, heads, head_conv=head_conv, dcn=True)


def get_pose_net(num_layers, heads, head_conv):
    """arch 'res_N' (msra_resnet.py:275-280)."""
    block, layers = resnet_spec[num_layers]
    return PoseResNet(block, layers, heads, head_conv=head_conv, dcn=False)
