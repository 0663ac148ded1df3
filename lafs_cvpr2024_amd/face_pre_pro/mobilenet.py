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


class _BNActFn(torch.autograd.Function):
    """BatchNorm2d (training or eval) + optional ReLU / h-swish, forward and backward, on the fused HIP kernels (fp32 NCHW).
    Running statistics are updated in place by the forward kernel exactly as nn.BatchNorm2d does (num_batches_tracked by the
    caller)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, training, act):
        x = x.contiguous()
        N, C, H, W = x.shape
        dev = x.device
        stat = torch.empty(2 * C, device=dev, dtype=torch.float32)
        ws = torch.empty(2 * C, device=dev, dtype=torch.float32) if training else None
        y = torch.empty_like(x)
        call("lafs_bn_act_fwd_nchw", _p(x), _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps), float(momentum),
             1 if training else 0, N, C, H * W, act, _p(ws), _p(stat), _p(y))
        ctx.save_for_backward(x, gamma, beta, stat)
        ctx.training, ctx.act = training, act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stat = ctx.saved_tensors
        N, C, H, W = x.shape
        dsum = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        dx = torch.empty_like(x)
        call("lafs_bn_act_bwd_nchw", _p(x), _p(dy.contiguous()), _p(stat), _p(gamma), _p(beta), 1 if ctx.training else 0, N, C, H * W,
             ctx.act, _p(dsum), _p(dx))
        return dx, dsum[C:], dsum[:C], None, None, None, None, None, None


_ACT_CODE = {nn.ReLU: _lib.ACT_RELU, nn.Hardswish: _lib.ACT_HSWISH, nn.Identity: _lib.ACT_NONE, type(None): _lib.ACT_NONE}


def bn_act(x, bn, act=None):
    """act(bn(x)) for a BatchNorm2d module and a ReLU / Hardswish / None.  Default: the stock modules (MIOpen's BatchNorm is
    already bandwidth-efficient: the fused HIP path -- one statistics pass + one apply pass, `lafs_bn_act_*_nchw` -- measured
    0.9 ms SLOWER per fine-tune step, 37.6 vs 36.7 ms).  LAFS_BN_FUSED=1 selects the fused kernels (kept correct by
    tests/test_gpu_kernels.py) for fp32 device tensors."""
    fused = (os.environ.get("LAFS_BN_FUSED") == "1" and x.is_cuda and x.dtype == torch.float32 and bn.weight.dtype == torch.float32
             and type(act) in _ACT_CODE and bn.track_running_stats and bn.affine)
    if not fused:
        y = bn(x)
        return act(y) if act is not None else y
    if bn.training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    momentum = 0.1 if bn.momentum is None else bn.momentum
    return _BNActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum, bn.training, _ACT_CODE[type(act)])


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
        c = self.conv                                      # same modules / state_dict keys as the reference's nn.Sequential
        y = bn_act(c[0](x), c[1], c[2])
        y = c[3](y)
        if isinstance(c[5], nn.Identity):
            y = bn_act(y, c[4], c[6])
        else:                                              # squeeze-excite sits between the BatchNorm and the activation
            y = c[6](c[5](bn_act(y, c[4])))
        y = bn_act(c[7](y), c[8])
        return x + y if self.residual else y


class _Stem(nn.Sequential):
    def __init__(self):
        super().__init__(nn.Conv2d(3, 16, 3, 2, 1, bias=False), nn.BatchNorm2d(16), nn.Hardswish(inplace=True))

    def forward(self, x):
        return bn_act(self[0](x), self[1], self[2])


class MobileNetV3_backbone(nn.Module):
    def __init__(self, n_class=1000, input_size=224, dropout=0.8, mode='small', width_mult=1.0):
        super().__init__()
        if mode != 'large' or width_mult != 1.0:
            raise NotImplementedError("the landmark CNN is MobileNetV3-large at width 1.0")
        layers = [_Stem()]
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
