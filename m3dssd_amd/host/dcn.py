"""DCNv2 / DCN modules with the reference's interface (model/DCNv2/dcn_v2.py:14-70,
model/DCNv2/dcn_v2_func.py:13-38), backed by m3d_dcn_v2_forward."""
import math

import torch
from torch import nn
from torch.nn.modules.utils import _pair

from . import ops


class DCNv2Function:
    """Callable with the reference's legacy instance-style convention:
    ``DCNv2Function(stride, padding, dilation, deformable_groups)(input, offset, mask, weight, bias)``.
    Inference only (the backward of the reference, dcn_v2_func.py:40-62, is out of scope)."""

    def __init__(self, stride, padding, dilation=1, deformable_groups=1):
        self.stride, self.padding, self.dilation, self.deformable_groups = stride, padding, dilation, deformable_groups

    def __call__(self, input, offset, mask, weight, bias):
        return self.forward(input, offset, mask, weight, bias)

    def forward(self, input, offset, mask, weight, bias):
        if not input.is_cuda:
            raise NotImplementedError
        if torch.is_grad_enabled() and any(t.requires_grad for t in (input, offset, mask)):
            raise NotImplementedError("m3dssd_amd implements the DCNv2 forward only (inference path)")
        with torch.no_grad():
            return ops.dcn_v2_forward(input, offset, mask, weight, bias, self.stride, self.padding, self.dilation,
                                      self.deformable_groups)


class DCNv2(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        fan = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        bound = 1.0 / math.sqrt(fan)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            self.bias.zero_()

    def forward(self, input, offset, mask):
        fn = DCNv2Function(self.stride, self.padding, self.dilation, self.deformable_groups)
        return fn(input, offset, mask, self.weight, self.bias)


class DCN(DCNv2):
    """DCNv2 whose offsets and mask come from its own zero-initialised conv (dcn_v2.py:44-70)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        kk = self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset_mask = nn.Conv2d(in_channels, deformable_groups * 3 * kk, kernel_size=self.kernel_size,
                                          stride=(stride, stride), padding=(padding, padding), bias=True)
        self.init_offset()

    def init_offset(self):
        with torch.no_grad():
            self.conv_offset_mask.weight.zero_()
            self.conv_offset_mask.bias.zero_()

    def forward(self, input):
        from .standalone import dcn_layer_forward
        return dcn_layer_forward(self, input)
