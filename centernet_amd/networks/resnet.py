"""ResNet trunks with an up-sampling path and CenterNet heads.

Two reference files are covered by one class:
  arch 'res_N'    : src/lib/models/networks/msra_resnet.py  (ConvTranspose up-path)
  arch 'resdcn_N' : src/lib/models/networks/resnet_dcn.py   (DCN + ConvTranspose up-path)
Parameter names/shapes follow those files exactly (conv1, bn1, layer{1..4}.{i}.conv{1,2,3},
layer*.0.downsample.{0,1}, deconv_layers.{i}, <head>.{0,2}).
"""
import torch.nn as nn

from ..dcn_v2 import DCN
from ..engine import PlannedModule
from .common import bn, conv, bilinear_upsample_init_, detection_head


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.stride, self.downsample = stride, downsample
        self.conv1, self.bn1 = conv(inplanes, planes, 3, stride), bn(planes)
        self.conv2, self.bn2 = conv(planes, planes, 3), bn(planes)

    def describe(self, pb, x, out_plain=False):
        # resnet_dcn.py:49-67
        res = x
        if self.downsample is not None:
            ds = self.downsample
            res = pb.conv(x, ds[0].weight, bn=ds[1], stride=ds[0].stride[0])
        out = pb.conv(x, self.conv1.weight, bn=self.bn1, relu=True, stride=self.stride, padding=1)
        return pb.conv(out, self.conv2.weight, bn=self.bn2, relu=True, residual=res, padding=1,
                       out_plain=out_plain)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        wide = planes * self.expansion
        self.stride, self.downsample = stride, downsample
        self.conv1, self.bn1 = conv(inplanes, planes, 1), bn(planes)
        self.conv2, self.bn2 = conv(planes, planes, 3, stride), bn(planes)
        self.conv3, self.bn3 = conv(planes, wide, 1), bn(wide)

    def describe(self, pb, x, out_plain=False):
        # resnet_dcn.py:88-108
        res = x
        if self.downsample is not None:
            ds = self.downsample
            res = pb.conv(x, ds[0].weight, bn=ds[1], stride=ds[0].stride[0])
        out = pb.conv(x, self.conv1.weight, bn=self.bn1, relu=True)
        out = pb.conv(out, self.conv2.weight, bn=self.bn2, relu=True, stride=self.stride, padding=1)
        return pb.conv(out, self.conv3.weight, bn=self.bn3, relu=True, residual=res,
                       out_plain=out_plain)


class PoseResNet(PlannedModule):
    STAGE_WIDTHS = (64, 128, 256, 512)

    def __init__(self, block, layers, heads, head_conv, dcn=True):
        super().__init__()
        self.heads, self.use_dcn = heads, dcn
        self.deconv_with_bias = False
        self.conv1, self.bn1 = conv(3, 64, 7, 2), bn(64)
        self.relu, self.maxpool = nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1)
        width = 64
        for stage, (planes, count) in enumerate(zip(self.STAGE_WIDTHS, layers), start=1):
            blocks, width = self._stage(block, width, planes, count, stride=1 if stage == 1 else 2)
            setattr(self, 'layer%d' % stage, blocks)
        up_widths = (256, 128, 64) if dcn else (256, 256, 256)
        self.deconv_layers = self._up_path(width, up_widths)
        # msra_resnet.py:133 creates the heads in sorted() order, resnet_dcn.py:155 in dict order;
        # only RNG-dependent initialisation differs, the names are the same
        for name in (heads if dcn else sorted(heads)):
            setattr(self, name, detection_head(up_widths[-1], head_conv, heads[name],
                                               is_heatmap='hm' in name, normal_init=dcn))

    @staticmethod
    def _stage(block, width_in, planes, count, stride):
        """One residual stage: the first block changes stride / width (with a 1x1 projection on
        the identity branch when needed), the others keep both."""
        width_out = planes * block.expansion
        project = None
        if stride != 1 or width_in != width_out:
            project = nn.Sequential(conv(width_in, width_out, 1, stride), bn(width_out))
        blocks = [block(width_in, planes, stride, project)]
        blocks += [block(width_out, planes) for _ in range(count - 1)]
        return nn.Sequential(*blocks), width_out

    def _up_path(self, width, up_widths):
        """Three x2 up-sampling steps: [DCN, BN, ReLU,] ConvTranspose 4x4/2, BN, ReLU."""
        mods = []
        for planes in up_widths:
            if self.use_dcn:
                mods += [DCN(width, planes, kernel_size=(3, 3), stride=1, padding=1, dilation=1,
                             deformable_groups=1), bn(planes), nn.ReLU(inplace=True)]
                width = planes
            up = nn.ConvTranspose2d(width, planes, 4, 2, 1, 0, bias=self.deconv_with_bias)
            if self.use_dcn:
                bilinear_upsample_init_(up)
            mods += [up, bn(planes), nn.ReLU(inplace=True)]
            width = planes
        return nn.Sequential(*mods)

    def describe(self, pb, x):
        # resnet_dcn.py:248-263 / msra_resnet.py forward
        x = pb.conv(x, self.conv1.weight, bn=self.bn1, relu=True, stride=2, padding=3,
                    pool=(self.maxpool.kernel_size, self.maxpool.stride, self.maxpool.padding))
        mods = list(self.deconv_layers)
        blocks = [blk for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for blk in layer]
        for blk in blocks:
            # the trunk's last layer feeds a deformable layer (plain floats) in the DCN variants
            x = blk.describe(pb, x, out_plain=blk is blocks[-1] and bool(mods) and isinstance(mods[0], DCN))
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, DCN):
                x = pb.dcn(x, m, bn=mods[i + 1], relu=True)
                i += 3
            elif isinstance(m, nn.ConvTranspose2d):
                # a deformable layer gathers plain floats: its producer writes them directly
                to_dcn = i + 3 < len(mods) and isinstance(mods[i + 3], DCN)
                x = pb.conv_transpose4x4s2(x, m.weight, bn=mods[i + 1], relu=True, out_plain=to_dcn)
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
