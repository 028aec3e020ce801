"""DLA-34 with deformable up-sampling (arch 'dla_34'), parameter container + plan.

Reference: src/lib/models/networks/pose_dla_dcn.py -- DLA (:224-306), Tree (:168-221),
Root (:147-165), BasicBlock (:31-62), DeformConv (:345-357), IDAUp (:360-386),
DLAUp (:390-413), DLASeg (:427-482).  State-dict names follow that file
(base.level2.tree1.conv1.weight, dla_up.ida_0.proj_1.conv.conv_offset_mask.bias,
ida_up.node_2.actf.0.running_var, hm.0.weight, ...).
"""
import os

import numpy as np
from torch import nn

from ..dcn_v2 import DCN
from ..engine import PlannedModule
from .common import bn, conv, bilinear_upsample_init_, detection_head


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, dilation=1):
        super().__init__()
        self.stride, self.dilation = stride, dilation
        self.conv1, self.bn1 = conv(inplanes, planes, 3, stride, dilation=dilation), bn(planes)
        self.conv2, self.bn2 = conv(planes, planes, 3, dilation=dilation), bn(planes)

    def describe(self, pb, x, residual=None, out=None):
        # pose_dla_dcn.py:45-62.  ``out``: the block's result goes straight into this Act (a channel
        # slice of the Root's concatenation buffer) where the kernel allows it
        if residual is None:
            residual = x
        mid = pb.conv(x, self.conv1.weight, bn=self.bn1, relu=True, stride=self.stride,
                      padding=self.dilation, dilation=self.dilation)
        return pb.conv(mid, self.conv2.weight, bn=self.bn2, relu=True, residual=residual,
                       padding=self.dilation, dilation=self.dilation, out=out)


class Root(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, residual):
        super().__init__()
        self.residual = residual
        self.conv = conv(in_channels, out_channels, 1, pad=(kernel_size - 1) // 2)
        self.bn = bn(out_channels)

    def describe(self, pb, *children, out_plain=False, into=None):
        # pose_dla_dcn.py:157-165: conv1x1(cat(children)) + BN (+ children[0]) + ReLU.  ``into``: the
        # concatenation buffer the Tree allocated up front (members written in place are not copied)
        x = pb.concat(list(children), into=into)
        return pb.conv(x, self.conv.weight, bn=self.bn, relu=True,
                       residual=children[0] if self.residual else None,
                       padding=self.conv.padding[0], out_plain=out_plain)


class Tree(nn.Module):
    """A node of the aggregation tree: two sub-trees (blocks at depth 1) whose outputs, plus the
    outputs handed down by the ancestors, are merged by a Root."""

    def __init__(self, levels, block, in_channels, out_channels, stride=1, level_root=False,
                 root_dim=0, root_kernel_size=1, dilation=1, root_residual=False):
        super().__init__()
        self.levels, self.stride, self.level_root = levels, stride, level_root
        root_dim = root_dim or 2 * out_channels
        if level_root:
            root_dim += in_channels
        self.root_dim = root_dim
        if levels == 1:
            self.tree1 = block(in_channels, out_channels, stride, dilation=dilation)
            self.tree2 = block(out_channels, out_channels, 1, dilation=dilation)
            self.root = Root(root_dim, out_channels, root_kernel_size, root_residual)
        else:
            shared = dict(root_kernel_size=root_kernel_size, dilation=dilation,
                          root_residual=root_residual)
            self.tree1 = Tree(levels - 1, block, in_channels, out_channels, stride, root_dim=0,
                              **shared)
            self.tree2 = Tree(levels - 1, block, out_channels, out_channels,
                              root_dim=root_dim + out_channels, **shared)
        self.downsample = nn.MaxPool2d(stride, stride=stride) if stride > 1 else None
        self.project = None
        if in_channels != out_channels:
            self.project = nn.Sequential(conv(in_channels, out_channels, 1), bn(out_channels))

    def describe(self, pb, x, residual=None, children=None, out_plain=False):
        # pose_dla_dcn.py:206-221.  ``out_plain``: the stage's result is written as plain floats
        # (it feeds the deformable layers and skip adds of the up-sampling pyramid)
        children = [] if children is None else children
        # a leaf tree knows its Root's concatenation (x2, x1, handed-down children, bottom) before any
        # member exists: the buffer is allocated first and tree1 / tree2 / the max-pool write their
        # results into their slices (Root.forward's torch.cat, pose_dla_dcn.py:159, without the copies)
        buf = slots = None
        # (not with a residual Root, DLA(residual_root=True): its 1x1 convolution would take children[0]
        # -- a slice of the buffer, at the buffer's pitch -- as residual, and the 1x1 kernels want the
        # residual at the output's pitch; those trees keep the copying concatenation)
        if self.levels == 1 and self.downsample is not None and not self.root.residual:
            oc = self.tree1.conv2.weight.shape[0]
            widths = [oc, oc] + [c.C for c in children] + ([x.C] if self.level_root else [])
            Ho, Wo = x.H // self.stride, x.W // self.stride
            if x.H % self.stride == 0 and x.W % self.stride == 0:
                buf, slots = pb.concat_buffer(x.B, Ho, Wo, widths)
        pooled_out = slots[-1] if (slots is not None and self.level_root) else None
        bottom = pb.maxpool(x, self.stride, self.stride, 0, out=pooled_out, to_s=True) \
            if self.downsample is not None else x
        residual = pb.conv(bottom, self.project[0].weight, bn=self.project[1]) \
            if self.project is not None else bottom
        if self.level_root:
            children.append(bottom)
        if self.levels == 1:
            x1 = self.tree1.describe(pb, x, residual, out=slots[1] if slots is not None else None)
            x2 = self.tree2.describe(pb, x1, out=slots[0] if slots is not None else None)
            return self.root.describe(pb, x2, x1, *children, out_plain=out_plain, into=buf)
        x1 = self.tree1.describe(pb, x, residual)
        children.append(x1)
        return self.tree2.describe(pb, x1, children=children, out_plain=out_plain)


class DLA(nn.Module):
    def __init__(self, levels, channels, block=BasicBlock, residual_root=False, with_fc=True):
        super().__init__()
        self.channels = channels
        self.base_layer = nn.Sequential(conv(3, channels[0], 7), bn(channels[0]),
                                        nn.ReLU(inplace=True))
        self.level0 = self._plain_level(channels[0], channels[0], levels[0], stride=1)
        self.level1 = self._plain_level(channels[0], channels[1], levels[1], stride=2)
        for lv in range(2, 6):  # the aggregation stages: every one halves the resolution
            setattr(self, 'level%d' % lv,
                    Tree(levels[lv], block, channels[lv - 1], channels[lv], 2,
                         level_root=lv > 2, root_residual=residual_root))
        if with_fc:
            # the reference's pretrained loader attaches the ImageNet classifier
            # (pose_dla_dcn.py:294-305), so zoo checkpoints carry base.fc.*; unused in forward
            self.fc = conv(channels[-1], 1000, 1, bias=True)

    @staticmethod
    def _plain_level(cin, cout, count, stride):
        """``count`` x (conv3x3, BN, ReLU); only the first conv changes stride / width."""
        mods = []
        for i in range(count):
            mods += [conv(cin if i == 0 else cout, cout, 3, stride if i == 0 else 1), bn(cout),
                     nn.ReLU(inplace=True)]
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
                # stage outputs feed the next stage (3x3 / s2, max-pool, projection: f32s readers) AND
                # the deformable layers / skip adds of the pyramid (plain readers): CN_DLA_LEVEL_PLAIN=1
                # writes them plain (round 3), the default f32s with ONE plain copy for the pyramid
                x = level.describe(pb, x, out_plain=os.environ.get("CN_DLA_LEVEL_PLAIN", "0") == "1")
            y.append(x)
        return y


def dla34(with_fc=True):
    return DLA([1, 1, 1, 2, 2, 1], [16, 32, 64, 128, 256, 512], block=BasicBlock, with_fc=with_fc)


class DeformConv(nn.Module):
    def __init__(self, chi, cho):
        super().__init__()
        self.actf = nn.Sequential(bn(cho), nn.ReLU(inplace=True))
        self.conv = DCN(chi, cho, kernel_size=(3, 3), stride=1, padding=1, dilation=1,
                        deformable_groups=1)

    def describe(self, pb, x):
        # pose_dla_dcn.py:354-357: DCN -> BN -> ReLU, one fused launch (+ the offset conv).
        # Everything downstream in the up-sampling pyramid reads plain floats (the deformable
        # gather, the depthwise transposed conv and its skip add), so the result is written plain.
        return pb.dcn(x, self.conv, bn=self.actf[0], relu=True, out_plain=True)


class IDAUp(nn.Module):
    """Iterative deep aggregation: every deeper level is projected to ``o`` channels (DCN),
    up-sampled by its factor (depthwise bilinear transposed conv), added to the level above and
    refined (DCN)."""

    def __init__(self, o, channels, up_f):
        super().__init__()
        for i, (c, factor) in enumerate(zip(channels, up_f)):
            if i == 0:
                continue  # the shallowest level is the aggregation target itself
            f = int(factor)
            up = nn.ConvTranspose2d(o, o, 2 * f, stride=f, padding=f // 2, output_padding=0,
                                    groups=o, bias=False)
            setattr(self, 'proj_%d' % i, DeformConv(c, o))
            setattr(self, 'up_%d' % i, bilinear_upsample_init_(up))
            setattr(self, 'node_%d' % i, DeformConv(o, o))

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
        self.channels = channels
        widths = list(channels)
        inputs = list(channels if in_channels is None else in_channels)
        rel = np.array(scales, dtype=int)
        # from the deepest pair upwards: ida_i aggregates levels [j:] into level j, after which
        # those levels all carry widths[j] channels at level j's resolution
        for i, j in enumerate(range(len(widths) - 2, -1, -1)):
            setattr(self, 'ida_%d' % i, IDAUp(widths[j], inputs[j:], rel[j:] // rel[j]))
            rel[j + 1:] = rel[j]
            inputs[j + 1:] = [widths[j]] * (len(widths) - j - 1)

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
        if down_ratio not in (2, 4, 8, 16):
            raise ValueError("down_ratio must be 2, 4, 8 or 16")
        if base_name != 'dla34':
            raise KeyError("only DLA-34 is on the MI355X hot path")
        self.heads = heads
        self.first_level, self.last_level = int(np.log2(down_ratio)), last_level
        self.base = dla34(with_fc=with_fc)
        used = self.base.channels[self.first_level:]
        self.dla_up = DLAUp(self.first_level, used, [2 ** i for i in range(len(used))])
        fused = self.base.channels[self.first_level:self.last_level]
        self.ida_up = IDAUp(out_channel or used[0], fused, [2 ** i for i in range(len(fused))])
        for name, classes in heads.items():
            setattr(self, name, detection_head(used[0], head_conv, classes, final_kernel,
                                               is_heatmap='hm' in name))

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
