"""Shared builders of the network parameter containers.

The ``nn.Module`` trees under this package exist to HOLD parameters under the reference's
state-dict names (so zoo checkpoints load) and to describe themselves to the plan builder; the
helpers here create the recurring pieces -- BatchNorm with the reference's momentum, the
``conv3x3 -> ReLU -> conv1x1`` detection head with its bias initialisation, the bilinear
initialisation of transposed convolutions."""
import numpy as np
import torch
from torch import nn

BN_MOMENTUM = 0.1


def bn(channels):
    return nn.BatchNorm2d(channels, momentum=BN_MOMENTUM)


def conv(cin, cout, k, stride=1, pad=None, dilation=1, bias=False):
    pad = dilation * (k // 2) if pad is None else pad
    return nn.Conv2d(cin, cout, k, stride, pad, dilation, bias=bias)


def bilinear_upsample_init_(up):
    """Bilinear interpolation taps in input channel 0 of every output group of a transposed
    convolution, all other taps left as initialised (the reference's ``fill_up_weights``:
    resnet_dcn.py:110-119, pose_dla_dcn.py:327-336)."""
    k = up.weight.shape[2]
    f = int(np.ceil(k / 2))
    centre = (2 * f - 1 - f % 2) / (2.0 * f)
    ramp = 1.0 - np.abs(np.arange(k) / f - centre)
    kernel = torch.from_numpy(np.outer(ramp, ramp).astype(np.float32))
    with torch.no_grad():
        up.weight[:, 0] = kernel.to(up.weight.dtype)
    return up


def detection_head(cin, head_conv, classes, final_kernel=1, is_heatmap=False, normal_init=False):
    """``Sequential(conv3x3(cin, head_conv), ReLU, conv(head_conv, classes, final_kernel))`` or a
    single conv when ``head_conv == 0`` (resnet_dcn.py:155-177, pose_dla_dcn.py:446-468).
    Heat-map heads start with bias -2.19 on the last layer (sigmoid ~ 0.1), the others with
    zero bias (and N(0, 0.001) weights in the resnet_dcn variant)."""
    last = conv(head_conv if head_conv > 0 else cin, classes, final_kernel, bias=True)
    if head_conv > 0:
        head = nn.Sequential(conv(cin, head_conv, 3, bias=True), nn.ReLU(inplace=True), last)
    else:
        head = last
    if is_heatmap:
        nn.init.constant_(last.bias, -2.19)
    else:
        for m in head.modules():
            if isinstance(m, nn.Conv2d):
                if normal_init:
                    nn.init.normal_(m.weight, std=0.001)
                nn.init.constant_(m.bias, 0)
    return head
