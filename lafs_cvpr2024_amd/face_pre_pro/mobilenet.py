"""MobileNetV3-large trunk used as the landmark regressor "stn" (reference face_pre_pro/mobilenet.py:224-313):
3x112x112 -> 160x4x4.  Table-driven re-implementation with the reference's ``state_dict`` key layout
(``features.0.{0,1}``, ``features.{i}.conv.{0,1,3,4,5.fc.{0,2},7,8}``).  It runs on stock PyTorch-ROCm (MIOpen depthwise
convolutions): north_star lists no HIP kernel for it and it is frozen / in eval mode on the LAFS path."""
import os

import torch
import torch.nn as nn

from .. import _lib
from ..ops import _p, call

# kernel, expansion, out channels, squeeze-excite, non-linearity, stride   (MobileNetV3-large, Howard et al. Table 1)
_LARGE = ((3, 16, 16, 0, "RE", 1), (3, 64, 24, 0, "RE", 2), (3, 72, 24, 0, "RE", 1), (5, 72, 40, 1, "RE", 2),
          (5, 120, 40, 1, "RE", 1), (5, 120, 40, 1, "RE", 1), (3, 240, 80, 0, "HS", 2), (3, 200, 80, 0, "HS", 1),
          (3, 184, 80, 0, "HS", 1), (3, 184, 80, 0, "HS", 1), (3, 480, 112, 1, "HS", 1), (3, 672, 112, 1, "HS", 1),
          (5, 672, 160, 1, "HS", 2), (5, 960, 160, 1, "HS", 1), (5, 960, 160, 1, "HS", 1))


class _DepthwiseFn(torch.autograd.Function):
    """Depthwise k x k convolution (pad (k-1)//2, no bias) and both gradients on the HIP kernels (fp32 NCHW)."""

    @staticmethod
    def forward(ctx, x, w, stride):
        x = x.contiguous(); w = w.contiguous()
        N, C, H, W = x.shape
        k = w.shape[-1]
        Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
        y = torch.empty(N, C, Ho, Wo, device=x.device, dtype=torch.float32)
        call("lafs_dwconv_nchw_fwd", _p(x), _p(w), N, C, H, W, k, stride, _p(y))
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, C, H, W = x.shape
        k, stride = w.shape[-1], ctx.stride
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            call("lafs_dwconv_nchw_bwd_data", _p(dy), _p(w), N, C, H, W, k, stride, _p(dx))
        if ctx.needs_input_grad[1]:
            dw = torch.zeros_like(w)
            call("lafs_dwconv_nchw_bwd_weight", _p(x), _p(dy), N, C, H, W, k, stride, _p(dw))
        return dx, dw, None


class DepthwiseConv2d(nn.Conv2d):
    """nn.Conv2d(C, C, k, stride, (k-1)//2, groups=C, bias=False) with the same parameters / state_dict key, whose forward and
    backward on fp32 device tensors are the hand-written depthwise kernels (MIOpen has no tuned solver for these shapes on this
    image: its fall-backs cost ~14 ms of a 50 ms fine-tune step).  Other inputs (CPU, bf16) take the stock path."""

    def __init__(self, channels, k, stride):
        super().__init__(channels, channels, k, stride, (k - 1) // 2, groups=channels, bias=False)

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and self.weight.dtype == torch.float32 and self.kernel_size[0] in (3, 5) \
                and self.stride[0] in (1, 2) and os.environ.get("LAFS_DW_STOCK") != "1":      # env switch: A/B against MIOpen
            return _DepthwiseFn.apply(x, self.weight, self.stride[0])
        return super().forward(x)


def _act(kind):
    return nn.ReLU(inplace=True) if kind == "RE" else nn.Hardswish(inplace=True)      # hswish = x*relu6(x+3)/6


class _SqueezeExcite(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(ch, ch // 4, bias=False), nn.ReLU(inplace=True), nn.Linear(ch // 4, ch, bias=False),
                                nn.Hardsigmoid(inplace=True))                           # relu6(x+3)/6

    def forward(self, x):
        w = self.fc(x.mean(dim=(2, 3)))
        return x * w[:, :, None, None]


class _InvertedResidual(nn.Module):
    def __init__(self, cin, cout, k, stride, cexp, se, nl):
        super().__init__()
        self.residual = stride == 1 and cin == cout
        self.conv = nn.Sequential(
            nn.Conv2d(cin, cexp, 1, bias=False), nn.BatchNorm2d(cexp), _act(nl),
            DepthwiseConv2d(cexp, k, stride), nn.BatchNorm2d(cexp),
            _SqueezeExcite(cexp) if se else nn.Identity(), _act(nl),
            nn.Conv2d(cexp, cout, 1, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = self.conv(x)
        return x + y if self.residual else y


class MobileNetV3_backbone(nn.Module):
    def __init__(self, n_class=1000, input_size=224, dropout=0.8, mode='small', width_mult=1.0):
        super().__init__()
        if mode != 'large' or width_mult != 1.0:
            raise NotImplementedError("the landmark CNN is MobileNetV3-large at width 1.0")
        layers = [nn.Sequential(nn.Conv2d(3, 16, 3, 2, 1, bias=False), nn.BatchNorm2d(16), nn.Hardswish(inplace=True))]
        cin = 16
        for k, cexp, cout, se, nl, s in _LARGE:
            layers.append(_InvertedResidual(cin, cout, k, s, cexp, bool(se), nl))
            cin = cout
        self.features = nn.Sequential(*layers)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight); nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)

    def forward(self, x):
        return self.features(x)
