"""MobileNetV3-large trunk used as the landmark regressor "stn" (reference face_pre_pro/mobilenet.py:224-313):
3x112x112 -> 160x4x4.  Table-driven re-implementation with the reference's ``state_dict`` key layout
(``features.0.{0,1}``, ``features.{i}.conv.{0,1,3,4,5.fc.{0,2},7,8}``).  It runs on stock PyTorch-ROCm (MIOpen depthwise
convolutions): north_star lists no HIP kernel for it and it is frozen / in eval mode on the LAFS path."""
import torch.nn as nn

# kernel, expansion, out channels, squeeze-excite, non-linearity, stride   (MobileNetV3-large, Howard et al. Table 1)
_LARGE = ((3, 16, 16, 0, "RE", 1), (3, 64, 24, 0, "RE", 2), (3, 72, 24, 0, "RE", 1), (5, 72, 40, 1, "RE", 2),
          (5, 120, 40, 1, "RE", 1), (5, 120, 40, 1, "RE", 1), (3, 240, 80, 0, "HS", 2), (3, 200, 80, 0, "HS", 1),
          (3, 184, 80, 0, "HS", 1), (3, 184, 80, 0, "HS", 1), (3, 480, 112, 1, "HS", 1), (3, 672, 112, 1, "HS", 1),
          (5, 672, 160, 1, "HS", 2), (5, 960, 160, 1, "HS", 1), (5, 960, 160, 1, "HS", 1))


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
            nn.Conv2d(cexp, cexp, k, stride, (k - 1) // 2, groups=cexp, bias=False), nn.BatchNorm2d(cexp),
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
