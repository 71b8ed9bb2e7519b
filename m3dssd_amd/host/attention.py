"""ANAB asymmetric non-local block with the reference's interface (model/module/attention.py:120-216)."""
from torch import nn


class PAPAModule(nn.Module):
    """Parameter-free; kept so that ``ANAB.key_papa`` / ``value_papa`` exist like in the reference."""

    def __init__(self, sizes=(1, 4, 8), dimension=2):
        super().__init__()
        if dimension != 2:
            raise NotImplementedError
        self.sizes = tuple(sizes)


class ANAB(nn.Module):
    def __init__(self, ch, num_psp, psp_size=[1, 4, 8, 16], with_atten=True):
        super().__init__()
        if not with_atten or tuple(psp_size) != (1, 4, 8, 16):
            raise NotImplementedError("the HIP ANAB path implements with_atten=True, psp_size=[1,4,8,16]")
        self.inch = self.outch = ch
        self.key_num = sum(i ** 2 for i in psp_size)
        self.key_ch = self.key_num // 2
        self.with_atten = with_atten
        self.value_conv = nn.Conv2d(ch, ch, kernel_size=1, bias=False)
        self.spatial_conv = nn.Conv2d(ch, len(psp_size), kernel_size=1, bias=False)
        self.key_conv = nn.Conv2d(ch, self.key_ch, kernel_size=1, bias=False)
        self.query_conv = nn.Conv2d(ch, self.key_ch, kernel_size=1, bias=False)
        self.key_papa = PAPAModule(sizes=psp_size)
        self.value_papa = PAPAModule(sizes=psp_size)

    def forward(self, x):
        from .standalone import anab_forward
        return anab_forward(self, x)
