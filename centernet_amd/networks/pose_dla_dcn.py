"""DLA-34 with deformable up-sampling (arch 'dla_34'), parameter container + plan.

Reference: src/lib/models/networks/pose_dla_dcn.py -- DLA (:224-306), Tree (:168-221),
Root (:147-165), BasicBlock (:31-62), DeformConv (:345-357), IDAUp (:360-386),
DLAUp (:390-413), DLASeg (:427-482).  State-dict names follow that file
(base.level2.tree1.conv1.weight, dla_up.ida_0.proj_1.conv.conv_offset_mask.bias,
ida_up.node_2.actf.0.running_var, hm.0.weight, ...).
"""
import numpy as np
import torch
from torch import nn

from ..dcn_v2 import DCN
from ..engine import PlannedModule

BN_MOMENTUM = 0.1


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, dilation, dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.stride = stride
        self.dilation = dilation

    def describe(self, pb, x, residual=None):
        # pose_dla_dcn.py:45-62
        if residual is None:
            residual = x
        out = pb.conv(x, self.conv1.weight, bn=self.bn1, relu=True, stride=self.stride,
                      padding=self.dilation, dilation=self.dilation)
        return pb.conv(out, self.conv2.weight, bn=self.bn2, relu=True, residual=residual,
                       padding=self.dilation, dilation=self.dilation)


class Root(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, residual):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 1, stride=1, bias=False,
                              padding=(kernel_size - 1) // 2)
        self.bn = nn.BatchNorm2d(out_channels, momentum=BN_MOMENTUM)
        self.residual = residual

    def describe(self, pb, *children):
        # pose_dla_dcn.py:157-165: conv1x1(cat(children)) + BN (+ children[0]) + ReLU
        x = pb.concat(list(children))
        return pb.conv(x, self.conv.weight, bn=self.bn, relu=True,
                       residual=children[0] if self.residual else None,
                       padding=self.conv.padding[0])


class Tree(nn.Module):
    def __init__(self, levels, block, in_channels, out_channels, stride=1, level_root=False,
                 root_dim=0, root_kernel_size=1, dilation=1, root_residual=False):
        super().__init__()
        if root_dim == 0:
            root_dim = 2 * out_channels
        if level_root:
            root_dim += in_channels
        if levels == 1:
            self.tree1 = block(in_channels, out_channels, stride, dilation=dilation)
            self.tree2 = block(out_channels, out_channels, 1, dilation=dilation)
            self.root = Root(root_dim, out_channels, root_kernel_size, root_residual)
        else:
            self.tree1 = Tree(levels - 1, block, in_channels, out_channels, stride, root_dim=0,
                              root_kernel_size=root_kernel_size, dilation=dilation,
                              root_residual=root_residual)
            self.tree2 = Tree(levels - 1, block, out_channels, out_channels,
                              root_dim=root_dim + out_channels,
                              root_kernel_size=root_kernel_size, dilation=dilation,
                              root_residual=root_residual)
        self.level_root = level_root
        self.root_dim = root_dim
        self.downsample = None
        self.project = None
        self.levels = levels
        self.stride = stride
        if stride > 1:
            self.downsample = nn.MaxPool2d(stride, stride=stride)
        if in_channels != out_channels:
            self.project = nn.Sequential(
                nn.Conv2d(in_channels, out_channels, 1, 1, bias=False),
                nn.BatchNorm2d(out_channels, momentum=BN_MOMENTUM))

    def describe(self, pb, x, residual=None, children=None):
        # pose_dla_dcn.py:206-221
        children = [] if children is None else children
        bottom = pb.maxpool(x, self.stride, self.stride, 0) if self.downsample is not None else x
        residual = pb.conv(bottom, self.project[0].weight, bn=self.project[1]) \
            if self.project is not None else bottom
        if self.level_root:
            children.append(bottom)
        x1 = self.tree1.describe(pb, x, residual)
        if self.levels == 1:
            x2 = self.tree2.describe(pb, x1)
            return self.root.describe(pb, x2, x1, *children)
        children.append(x1)
        return self.tree2.describe(pb, x1, children=children)


class DLA(nn.Module):
    def __init__(self, levels, channels, block=BasicBlock, residual_root=False, with_fc=True):
        super().__init__()
        self.channels = channels
        self.base_layer = nn.Sequential(
            nn.Conv2d(3, channels[0], 7, 1, 3, bias=False),
            nn.BatchNorm2d(channels[0], momentum=BN_MOMENTUM), nn.ReLU(inplace=True))
        self.level0 = self._make_conv_level(channels[0], channels[0], levels[0])
        self.level1 = self._make_conv_level(channels[0], channels[1], levels[1], stride=2)
        self.level2 = Tree(levels[2], block, channels[1], channels[2], 2, level_root=False,
                           root_residual=residual_root)
        self.level3 = Tree(levels[3], block, channels[2], channels[3], 2, level_root=True,
                           root_residual=residual_root)
        self.level4 = Tree(levels[4], block, channels[3], channels[4], 2, level_root=True,
                           root_residual=residual_root)
        self.level5 = Tree(levels[5], block, channels[4], channels[5], 2, level_root=True,
                           root_residual=residual_root)
        if with_fc:
            # the reference's pretrained loader attaches the ImageNet classifier
            # (pose_dla_dcn.py:294-305), so zoo checkpoints carry base.fc.*; unused in forward
            self.fc = nn.Conv2d(channels[-1], 1000, 1, 1, 0, bias=True)

    @staticmethod
    def _make_conv_level(inplanes, planes, convs, stride=1, dilation=1):
        mods = []
        for i in range(convs):
            mods.extend([nn.Conv2d(inplanes, planes, 3, stride if i == 0 else 1, dilation,
                                   dilation, bias=False),
                         nn.BatchNorm2d(planes, momentum=BN_MOMENTUM), nn.ReLU(inplace=True)])
            inplanes = planes
        return nn.Sequential(*mods)

    def describe(self, pb, x):
        # pose_dla_dcn.py:280-286
        y = []
        bl = self.base_layer
        x = pb.conv(x, bl[0].weight, bn=bl[1], relu=True, stride=1, padding=3)
        for i in range(6):
            level = getattr(self, 'level{}'.format(i))
            if isinstance(level, nn.Sequential):
                mods = list(level)
                for j in range(0, len(mods), 3):
                    c = mods[j]
                    x = pb.conv(x, c.weight, bn=mods[j + 1], relu=True, stride=c.stride[0],
                                padding=c.padding[0], dilation=c.dilation[0])
            else:
                x = level.describe(pb, x)
            y.append(x)
        return y


def dla34(with_fc=True):
    return DLA([1, 1, 1, 2, 2, 1], [16, 32, 64, 128, 256, 512], block=BasicBlock, with_fc=with_fc)


class DeformConv(nn.Module):
    def __init__(self, chi, cho):
        super().__init__()
        self.actf = nn.Sequential(nn.BatchNorm2d(cho, momentum=BN_MOMENTUM), nn.ReLU(inplace=True))
        self.conv = DCN(chi, cho, kernel_size=(3, 3), stride=1, padding=1, dilation=1,
                        deformable_groups=1)

    def describe(self, pb, x):
        # pose_dla_dcn.py:354-357: DCN -> BN -> ReLU, one fused launch (+ the offset conv)
        return pb.dcn(x, self.conv, bn=self.actf[0], relu=True)


def _fill_up_weights(up):
    import math
    w = up.weight.data
    f = math.ceil(w.size(2) / 2)
    c = (2 * f - 1 - f % 2) / (2. * f)
    for i in range(w.size(2)):
        for j in range(w.size(3)):
            w[0, 0, i, j] = (1 - math.fabs(i / f - c)) * (1 - math.fabs(j / f - c))
    for ch in range(1, w.size(0)):
        w[ch, 0, :, :] = w[0, 0, :, :]


class IDAUp(nn.Module):
    def __init__(self, o, channels, up_f):
        super().__init__()
        for i in range(1, len(channels)):
            c = channels[i]
            f = int(up_f[i])
            setattr(self, 'proj_' + str(i), DeformConv(c, o))
            up = nn.ConvTranspose2d(o, o, f * 2, stride=f, padding=f // 2, output_padding=0,
                                    groups=o, bias=False)
            _fill_up_weights(up)
            setattr(self, 'up_' + str(i), up)
            setattr(self, 'node_' + str(i), DeformConv(o, o))

    def describe(self, pb, layers, startp, endp):
        # pose_dla_dcn.py:379-386; the "+ layers[i-1]" is fused into the up-sampling kernel
        for i in range(startp + 1, endp):
            up = getattr(self, 'up_' + str(i - startp))
            proj = getattr(self, 'proj_' + str(i - startp))
            node = getattr(self, 'node_' + str(i - startp))
            p = proj.describe(pb, layers[i])
            s = pb.dw_deconv(p, up.weight, up.stride[0], add=layers[i - 1])
            layers[i] = node.describe(pb, s)


class DLAUp(nn.Module):
    def __init__(self, startp, channels, scales, in_channels=None):
        super().__init__()
        self.startp = startp
        if in_channels is None:
            in_channels = channels
        self.channels = channels
        channels = list(channels)
        in_channels = list(in_channels)
        scales = np.array(scales, dtype=int)
        for i in range(len(channels) - 1):
            j = -i - 2
            setattr(self, 'ida_{}'.format(i),
                    IDAUp(channels[j], in_channels[j:], scales[j:] // scales[j]))
            scales[j + 1:] = scales[j]
            in_channels[j + 1:] = [channels[j] for _ in channels[j + 1:]]

    def describe(self, pb, layers):
        # pose_dla_dcn.py:407-413
        layers = list(layers)
        out = [layers[-1]]
        for i in range(len(layers) - self.startp - 1):
            ida = getattr(self, 'ida_{}'.format(i))
            ida.describe(pb, layers, len(layers) - i - 2, len(layers))
            out.insert(0, layers[-1])
        return out


class DLASeg(PlannedModule):
    def __init__(self, base_name, heads, down_ratio, final_kernel, last_level, head_conv,
                 out_channel=0, with_fc=True):
        super().__init__()
        assert down_ratio in [2, 4, 8, 16]
        assert base_name == 'dla34', "only DLA-34 is on the MI355X hot path"
        self.first_level = int(np.log2(down_ratio))
        self.last_level = last_level
        self.base = dla34(with_fc=with_fc)
        channels = self.base.channels
        scales = [2 ** i for i in range(len(channels[self.first_level:]))]
        self.dla_up = DLAUp(self.first_level, channels[self.first_level:], scales)
        if out_channel == 0:
            out_channel = channels[self.first_level]
        self.ida_up = IDAUp(out_channel, channels[self.first_level:self.last_level],
                            [2 ** i for i in range(self.last_level - self.first_level)])
        self.heads = heads
        for head in self.heads:
            classes = self.heads[head]
            if head_conv > 0:
                fc = nn.Sequential(
                    nn.Conv2d(channels[self.first_level], head_conv, 3, padding=1, bias=True),
                    nn.ReLU(inplace=True),
                    nn.Conv2d(head_conv, classes, final_kernel, stride=1,
                              padding=final_kernel // 2, bias=True))
                last = fc[-1]
            else:
                fc = nn.Conv2d(channels[self.first_level], classes, final_kernel, stride=1,
                               padding=final_kernel // 2, bias=True)
                last = fc
            if 'hm' in head:
                last.bias.data.fill_(-2.19)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d) and m.bias is not None:
                        nn.init.constant_(m.bias, 0)
            self.__setattr__(head, fc)

    def describe(self, pb, x):
        # pose_dla_dcn.py:470-482 (the .clone() there only protects x from in-place edits)
        x = self.base.describe(pb, x)
        x = self.dla_up.describe(pb, x)
        y = [x[i] for i in range(self.last_level - self.first_level)]
        self.ida_up.describe(pb, y, 0, len(y))
        return pb.heads(y[-1], {h: getattr(self, h) for h in self.heads})


def get_pose_net(num_layers, heads, head_conv=256, down_ratio=4):
    """arch 'dla_34' (pose_dla_dcn.py:485-492); no ImageNet download (offline image)."""
    return DLASeg('dla{}'.format(num_layers), heads, down_ratio=down_ratio, final_kernel=1,
                  last_level=5, head_conv=head_conv)
