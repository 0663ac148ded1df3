"""Inference engine of the FROZEN landmark CNN on the HIP kernels.

The LAFS step runs `landmarkcnn` (MobileNetV3-large trunk `stn` + `output_layer`, reference face_pre_pro/mobilenet.py:224-313,
face_pre_pro/ViT_face.py:1338-1344) in eval mode on 10*B images per step (lafs_train.py:262-269, 535-567).  On stock PyTorch
that is ~150 MIOpen / elementwise launches over fp32 NCHW activations and costs as much as the whole ViT step's forward.
Here the network is compiled once into a launch plan over NHWC bf16 activations:

  * BatchNorm (eval) is folded into the convolution weights and a bias;
  * channel counts are padded to multiples of 32 so that every 1x1 convolution is ONE `lafs_gemm_nt` call whose epilogue
    (LAFS_EPI_BF16_ACT) adds the folded bias and the residual and applies ReLU / h-swish / h-sigmoid;
  * the 3x3 stem, the depthwise 3x3 / 5x5 convolutions (+bias +activation), the squeeze-excite pooling and the
    excite-rescale(+activation) are bandwidth-bound kernels (`csrc/landmark_cnn.hip`), 16-byte accesses along the channel axis;
  * the squeeze-excite FCs and the final Linear(160, 2*n) are `lafs_gemm_nt` calls too (batch rows x channels).

fp32 accumulation everywhere; activations are rounded to bf16 between layers (theta moves by a fraction of a pixel, far
below the 5 px jitter the step adds on top, ViT_face.py:1361-1362).  Trainable use of the CNN (Part-fViT with_land=True in
train_largescale.py) stays on torch autograd: this engine has no backward.
"""
import torch
import torch.nn as nn

from . import _lib, ops
from .ops import _p, call

f32, bf16 = torch.float32, torch.bfloat16


def _pad32(c):
    return (c + 31) // 32 * 32


def _act_code(m):
    if isinstance(m, nn.ReLU):
        return _lib.ACT_RELU
    if isinstance(m, nn.Hardswish):
        return _lib.ACT_HSWISH
    if isinstance(m, nn.Hardsigmoid):
        return _lib.ACT_HSIGMOID
    if isinstance(m, nn.Identity):
        return _lib.ACT_NONE
    raise _lib.LafsHipError(f"unsupported activation {type(m).__name__}")


def _fold(conv, bn):
    """eval-mode BatchNorm folded into the preceding bias-free convolution: returns (weight fp64, bias fp64)."""
    w = conv.weight.detach().double()
    s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    b = bn.bias.detach().double() - bn.running_mean.detach().double() * s
    if conv.bias is not None:
        b = b + conv.bias.detach().double() * s
    return w * s.view(-1, 1, 1, 1), b


class HipLandmarkCNN:
    def __init__(self, module, device=None):
        """module: anything with `.stn` (MobileNetV3_backbone) and `.output_layer` = Sequential(Dropout, Linear)."""
        self.device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        dev = self.device
        feats = module.stn.features
        padw = lambda w2d, r, c: torch.nn.functional.pad(w2d, (0, c - w2d.shape[1], 0, r - w2d.shape[0]))
        padv = lambda v, n: torch.nn.functional.pad(v, (0, n - v.shape[0]))
        # stem: [16,3,3,3] -> [(c,ky,kx)][o]
        w, b = _fold(feats[0][0], feats[0][1])
        self.stem_w = w.permute(1, 2, 3, 0).reshape(27, 16).to(dev, f32).contiguous()
        self.stem_b = b.to(dev, f32).contiguous()
        self.stem_act = _act_code(feats[0][2])
        self.c0 = _pad32(16)
        self.blocks = []
        cin = 16
        for blk in feats[1:]:
            cv = blk.conv
            cexp, cout = cv[0].out_channels, cv[7].out_channels
            k, stride = cv[3].kernel_size[0], cv[3].stride[0]
            pi, pe, po = _pad32(cin), _pad32(cexp), _pad32(cout)
            L = dict(cin=cin, cexp=cexp, cout=cout, pi=pi, pe=pe, po=po, k=k, stride=stride, residual=bool(blk.residual),
                     act=_act_code(cv[2]))
            w, b = _fold(cv[0], cv[1])
            L["w_exp"] = padw(w.view(cexp, cin), pe, pi).to(dev, bf16).contiguous()
            L["b_exp"] = padv(b, pe).to(dev, f32).contiguous()
            w, b = _fold(cv[3], cv[4])
            L["w_dw"] = padw(w.view(cexp, k * k).t(), k * k, pe).to(dev, f32).contiguous()        # [k*k][C]
            L["b_dw"] = padv(b, pe).to(dev, f32).contiguous()
            se = cv[5]
            if not isinstance(se, nn.Identity):
                h = se.fc[0].out_features
                ph = _pad32(h)
                L["se"] = dict(ph=ph, w1=padw(se.fc[0].weight.detach().double(), ph, pe).to(dev, bf16).contiguous(),
                               w2=padw(se.fc[2].weight.detach().double(), pe, ph).to(dev, bf16).contiguous(),
                               act1=_act_code(se.fc[1]), act2=_act_code(se.fc[3]))
            w, b = _fold(cv[7], cv[8])
            L["w_proj"] = padw(w.view(cout, cexp), po, pe).to(dev, bf16).contiguous()
            L["b_proj"] = padv(b, po).to(dev, f32).contiguous()
            self.blocks.append(L)
            cin = cout
        self.c_last, self.p_last = cin, _pad32(cin)
        lin = module.output_layer[1]
        self.n_out = lin.out_features
        self.w_head = padw(lin.weight.detach().double(), lin.out_features, self.p_last).to(dev, bf16).contiguous()
        self.b_head = lin.bias.detach().to(dev, f32).contiguous()
        self._plans = {}

    # ------------------------------------------------------------------ buffers for one batch size / resolution
    def _plan(self, N, S):
        key = (N, S)
        if key in self._plans:
            return self._plans[key]
        dev = self.device
        H = S // 2
        bufs = dict(x0=torch.empty(N * H * H, self.c0, device=dev, dtype=bf16), layers=[])
        for L in self.blocks:
            Ho = (H + L["stride"] - 1) // L["stride"]
            d = dict(H=H, Ho=Ho, e=torch.empty(N * H * H, L["pe"], device=dev, dtype=bf16),
                     d=torch.empty(N * Ho * Ho, L["pe"], device=dev, dtype=bf16),
                     y=torch.empty(N * Ho * Ho, L["po"], device=dev, dtype=bf16))
            if "se" in L:
                d["pool"] = torch.empty(N, L["pe"], device=dev, dtype=bf16)
                d["hid"] = torch.empty(N, L["se"]["ph"], device=dev, dtype=bf16)
                d["gate"] = torch.empty(N, L["pe"], device=dev, dtype=bf16)
            bufs["layers"].append(d)
            H = Ho
        bufs["Hlast"] = H
        bufs["feat"] = torch.empty(N, self.p_last, device=dev, dtype=bf16)
        bufs["t"] = torch.empty(N, self.n_out, device=dev, dtype=f32)
        self._plans[key] = bufs
        return bufs

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x):
        """x f32 NCHW [N,3,S,S] -> raw regressor output f32 [N, n_out] (before the min-max scaling)."""
        if not x.is_cuda or x.dtype != f32 or x.dim() != 4 or x.shape[1] != 3:
            raise _lib.LafsHipError("HipLandmarkCNN.forward expects a float32 NCHW device tensor with 3 channels")
        x = x.contiguous()
        N, S = x.shape[0], x.shape[-1]
        P = self._plan(N, S)
        call("lafs_cnn_stem", _p(x), _p(self.stem_w), _p(self.stem_b), N, S, self.stem_act, _p(P["x0"]), self.c0)
        cur = P["x0"]
        for L, B in zip(self.blocks, P["layers"]):
            H, Ho = B["H"], B["Ho"]
            ops.gemm_nt(cur, L["w_exp"], _lib.EPI_BF16_ACT, bias=L["b_exp"], out=B["e"], act=L["act"])
            se = L.get("se")
            call("lafs_cnn_dwconv", _p(B["e"]), _p(L["w_dw"]), _p(L["b_dw"]), N, H, H, L["pe"], L["k"], L["stride"],
                 -1 if se else L["act"], _p(B["d"]))
            if se:
                call("lafs_cnn_pool", _p(B["d"]), N, Ho * Ho, L["pe"], _p(B["pool"]), L["pe"])
                ops.gemm_nt(B["pool"], se["w1"], _lib.EPI_BF16_ACT, out=B["hid"], act=se["act1"])
                ops.gemm_nt(B["hid"], se["w2"], _lib.EPI_BF16_ACT, out=B["gate"], act=se["act2"])
                call("lafs_cnn_scale_act", _p(B["d"]), _p(B["gate"]), L["pe"], N, Ho * Ho, L["pe"], L["act"])
            ops.gemm_nt(B["d"], L["w_proj"], _lib.EPI_BF16_ACT, bias=L["b_proj"], out=B["y"], aux=cur if L["residual"] else None,
                        act=_lib.ACT_NONE)
            cur = B["y"]
        Hl = P["Hlast"]
        call("lafs_cnn_pool", _p(cur), N, Hl * Hl, self.p_last, _p(P["feat"]), self.p_last)
        ops.gemm_nt(P["feat"], self.w_head, _lib.EPI_F32, bias=self.b_head, out=P["t"])
        return P["t"]

    __call__ = forward
