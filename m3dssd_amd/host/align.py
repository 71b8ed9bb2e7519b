"""center_align / shape_align with the reference's interface (model/module/feturealign_mgpu.py:7-208)."""
import torch
from torch import nn
from torch.nn.modules.utils import _pair

from .dcn import DCNv2


class center_align(nn.Module):
    def __init__(self, ch, anchors, xy_mean, xy_std, feat_stride, feat_size, kernel_size=1, k=1, thresh=0.5):
        super().__init__()
        if k != 1:
            raise NotImplementedError("only k=1 (arg-max anchor) is implemented, as instantiated by RPN")
        self.ch, self.kernel_size, self.k = ch, _pair(kernel_size), k
        self.anchors = torch.as_tensor(anchors).clone().float().detach().cpu()
        self.num_anchors = self.anchors.shape[0]
        self.feat_stride, self.thresh, self.feat_size = feat_stride, thresh, feat_size
        self.xy_mean = torch.as_tensor(xy_mean, dtype=torch.float)
        self.xy_std = torch.as_tensor(xy_std, dtype=torch.float)
        self.anchors_w = ((self.anchors[:, 2] - self.anchors[:, 0]) / feat_stride).view(1, -1, 1, 1)
        self.anchors_h = ((self.anchors[:, 3] - self.anchors[:, 1]) / feat_stride).view(1, -1, 1, 1)
        self.align = DCNv2(ch, ch, self.kernel_size, 1, kernel_size // 2, dilation=1, deformable_groups=1)

    def forward(self, x, bbox_x, bbox_y, prob):
        from .standalone import center_align_forward
        return center_align_forward(self, x, bbox_x, bbox_y, prob)


class shape_align(nn.Module):
    def __init__(self, ch, anchors, feat_stride, feat_size, kernel_size=3, k=1, thresh=0.5):
        super().__init__()
        if k != 1:
            raise NotImplementedError("only k=1 (arg-max anchor) is implemented, as instantiated by RPN")
        self.ch, self.kernel_size, self.k = ch, _pair(kernel_size), k
        self.anchors = torch.as_tensor(anchors).clone().float().detach().cpu()
        self.num_anchors = self.anchors.shape[0]
        self.feat_stride, self.feat_size, self.thresh = feat_stride, feat_size, thresh
        kh, kw = self.kernel_size
        aw = self.anchors[:, 2] - self.anchors[:, 0]
        ah = self.anchors[:, 3] - self.anchors[:, 1]
        h_step, w_step = ah / feat_stride / kh, aw / feat_stride / kw
        tab = torch.zeros(self.num_anchors, 2 * kh * kw)
        for i in range(kh):
            for j in range(kw):
                tab[:, 2 * (i * kw + j)] = (h_step - 1) * (i - kh / 2 + 0.5)
                tab[:, 2 * (i * kw + j) + 1] = (w_step - 1) * (j - kw / 2 + 0.5)
        self.offset_table = tab                      # [A, 2*kh*kw]; the reference tiles it over H x W (:126-136)
        self.align = DCNv2(ch, ch, self.kernel_size, 1, kernel_size // 2, 1, deformable_groups=1)
        self.proj = nn.Conv2d(ch * 2, ch, 1, bias=False)   # present in the state_dict, unused by forward (:145,205-208)

    def forward(self, x, prob):
        from .standalone import shape_align_forward
        return shape_align_forward(self, x, prob)
