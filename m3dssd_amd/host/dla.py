"""DLA-34 backbone + DCN up-sampling modules with the reference's interface and state_dict keys
(model/pose_dla_dcn.py: BasicBlock :93-121, Root :251-269, Tree :272-327, DLA :330-397,
dla34 :419-425, DeformConv :471-485, IDAUp :519-552, DLAUp :556-578, DLASeg :641-696).

The modules are parameter containers; the arithmetic runs in the HIP engine
(m3dssd_amd/engine.py) -- ``DLASeg.forward`` executes the backbone part of the engine plan,
``DeformConv.forward`` the fused offset-conv + DCN + BN + LeakyReLU launches."""
import math

import numpy as np
import torch
from torch import nn

from .dcn import DCN

BN_MOMENTUM = 0.1


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=True, dilation=dilation)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.relu = nn.LeakyReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=True, dilation=dilation)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.stride = stride


class Root(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, residual):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 1, stride=1, bias=False, padding=(kernel_size - 1) // 2)
        self.bn = nn.BatchNorm2d(out_channels, momentum=BN_MOMENTUM)
        self.relu = nn.LeakyReLU(inplace=True)
        self.residual = residual


class Tree(nn.Module):
    def __init__(self, levels, block, in_channels, out_channels, stride=1, level_root=False, root_dim=0,
                 root_kernel_size=1, dilation=1, root_residual=False):
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
                              root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
            self.tree2 = Tree(levels - 1, block, out_channels, out_channels, root_dim=root_dim + out_channels,
                              root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
        self.level_root, self.root_dim, self.levels = level_root, root_dim, levels
        self.downsample = nn.MaxPool2d(stride, stride=stride) if stride > 1 else None
        self.project = None
        if in_channels != out_channels:
            self.project = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, stride=1, bias=False),
                                         nn.BatchNorm2d(out_channels, momentum=BN_MOMENTUM))


class DLA(nn.Module):
    def __init__(self, levels, channels, block=BasicBlock, residual_root=False):
        super().__init__()
        self.channels = channels
        self.base_layer = nn.Sequential(nn.Conv2d(3, channels[0], 7, stride=1, padding=3, bias=False),
                                        nn.BatchNorm2d(channels[0], momentum=BN_MOMENTUM), nn.LeakyReLU(inplace=True))
        self.level0 = self._conv_level(channels[0], channels[0], levels[0])
        self.level1 = self._conv_level(channels[0], channels[1], levels[1], stride=2)
        self.level2 = Tree(levels[2], block, channels[1], channels[2], 2, level_root=False, root_residual=residual_root)
        self.level3 = Tree(levels[3], block, channels[2], channels[3], 2, level_root=True, root_residual=residual_root)
        self.level4 = Tree(levels[4], block, channels[3], channels[4], 2, level_root=True, root_residual=residual_root)
        self.level5 = Tree(levels[5], block, channels[4], channels[5], 2, level_root=True, root_residual=residual_root)

    @staticmethod
    def _conv_level(inplanes, planes, convs, stride=1):
        mods = []
        for i in range(convs):
            mods += [nn.Conv2d(inplanes, planes, 3, stride=stride if i == 0 else 1, padding=1, bias=False),
                     nn.BatchNorm2d(planes, momentum=BN_MOMENTUM), nn.LeakyReLU(inplace=True)]
            inplanes = planes
        return nn.Sequential(*mods)


def dla34(pretrained=False, **kwargs):
    if pretrained:
        raise RuntimeError("no network access: load ImageNet weights yourself and call load_state_dict "
                           "(the reference downloads dla34-ba72cf86.pth, pose_dla_dcn.py:399-405)")
    return DLA([1, 1, 1, 2, 2, 1], [16, 32, 64, 128, 256, 512], block=BasicBlock, **kwargs)


def fill_up_weights(up):
    w = up.weight.data
    f = math.ceil(w.size(2) / 2)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    for i in range(w.size(2)):
        for j in range(w.size(3)):
            w[:, 0, i, j] = (1 - math.fabs(i / f - c)) * (1 - math.fabs(j / f - c))


class DeformConv(nn.Module):
    def __init__(self, chi, cho):
        super().__init__()
        self.actf = nn.Sequential(nn.BatchNorm2d(cho, momentum=BN_MOMENTUM), nn.LeakyReLU(inplace=True))
        self.conv = DCN(chi, cho, kernel_size=3, stride=1, padding=1, dilation=1, deformable_groups=1)

    def forward(self, x):
        from .standalone import deform_conv_forward
        return deform_conv_forward(self, x)


class IDAUp(nn.Module):
    def __init__(self, o, channels, up_f, conf):
        super().__init__()
        if not conf.ida_dcnv2:
            raise NotImplementedError("only the ida_dcnv2=True variant is on the M3DSSD hot path")
        self.out_channels = channels
        for i in range(1, len(channels)):
            f = int(up_f[i])
            setattr(self, "proj_%d" % i, DeformConv(channels[i], o))
            up = nn.ConvTranspose2d(o, o, f * 2, stride=f, padding=f // 2, output_padding=0, groups=o, bias=False)
            fill_up_weights(up)
            setattr(self, "up_%d" % i, up)
            setattr(self, "node_%d" % i, DeformConv(o, o))


class DLAUp(nn.Module):
    def __init__(self, startp, channels, scales, in_channels=None, conf=None):
        super().__init__()
        self.startp = startp
        in_channels = list(channels) if in_channels is None else in_channels
        self.channels = channels
        channels = list(channels)
        scales = np.array(scales, dtype=int)
        for i in range(len(channels) - 1):
            j = -i - 2
            setattr(self, "ida_%d" % i, IDAUp(channels[j], in_channels[j:], scales[j:] // scales[j], conf=conf))
            scales[j + 1:] = scales[j]
            in_channels[j + 1:] = [channels[j] for _ in channels[j + 1:]]


class DLASeg(nn.Module):
    def __init__(self, base_name, pretrained, down_ratio, final_kernel, last_level, head_conv, conf, out_channel=0):
        super().__init__()
        assert down_ratio in [2, 4, 8, 16]
        if base_name != "dla34":
            raise NotImplementedError("back_bone %r: this build accelerates the dla34 path only" % base_name)
        if down_ratio != 8 or last_level != 5:
            raise NotImplementedError("the HIP engine is laid out for feat_stride 8 / last_level 5")
        self.first_level = int(np.log2(down_ratio))
        self.last_level = last_level
        self.base = dla34(pretrained=pretrained)
        channels = self.base.channels
        scales = [2 ** i for i in range(len(channels[self.first_level:]))]
        self.dla_up = DLAUp(self.first_level, channels[self.first_level:], scales, conf=conf)
        if out_channel == 0:
            out_channel = channels[self.first_level]
        self.out_channels = out_channel
        self.ida_up = IDAUp(out_channel, channels[self.first_level:self.last_level],
                            [2 ** i for i in range(self.last_level - self.first_level)], conf)
        self._conf = conf
        self._engine = None

    def forward(self, x):
        from .standalone import dlaseg_forward
        return dlaseg_forward(self, x)
