"""Inference engine of the FROZEN landmark CNN on the HIP kernels.

The LAFS step runs `landmarkcnn` (MobileNetV3-large trunk `stn` + `output_layer`, reference face_pre_pro/mobilenet.py:224-313,
face_pre_pro/ViT_face.py:1338-1344) in eval mode on 10*B images per step (lafs_train.py:262-269, 535-567).  On stock PyTorch
that is ~150 MIOpen / elementwise launches over fp32 NCHW activations and costs as much as the whole ViT step's forward.
Here the network is compiled once into a launch plan over NHWC bf16 activations:

  * BatchNorm (eval) is folded into the convolution weights and a bias;
  * channel counts are padded to multiples of 32 so that every 1x1 convolution is ONE `lafs_gemm_nt` call whose epilogue
    (LAFS_EPI_BF16_ACT) adds the folded bias and the residual and applies ReLU / h-swish / h-sigmoid -- except on the 56 x 56 and
    28 x 28 maps, whose tensors are stored unpadded and read g pixels per GEMM row against a block-diagonal weight (`_plan`);
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


def _group(width):
    """Pixels per GEMM row for a tensor stored `width` channels wide: the smallest g with g * width a multiple of 32 (the GEMM's K
    granularity).  [M, width] read as [M / g, g * width] against a block-diagonal weight is the same convolution."""
    for g in (1, 2, 4):
        if (g * width) % 32 == 0:
            return g
    return None


PACK_MIN_SIDE = 28        # activations of H x H maps with H >= this are stored unpadded (the 56 x 56 / 28 x 28 stages: 2 of the step's 2.6 ms)


class HipLandmarkCNN:
    def __init__(self, module, device=None):
        """module: anything with `.stn` (MobileNetV3_backbone) and `.output_layer` = Sequential(Dropout, Linear)."""
        self.device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        dev = self.device
        feats = module.stn.features
        # stem: [16,3,3,3] -> [(c,ky,kx)][o]
        w, b = _fold(feats[0][0], feats[0][1])
        self.stem_w = w.permute(1, 2, 3, 0).reshape(27, 16).to(dev, f32).contiguous()
        self.stem_b = b.to(dev, f32).contiguous()
        self.stem_act = _act_code(feats[0][2])
        self.blocks = []
        cin = 16
        for blk in feats[1:]:                                  # folded fp64 weights; their device images depend on the plan's layout
            cv = blk.conv
            cexp, cout = cv[0].out_channels, cv[7].out_channels
            k, stride = cv[3].kernel_size[0], cv[3].stride[0]
            L = dict(cin=cin, cexp=cexp, cout=cout, k=k, stride=stride, residual=bool(blk.residual), act=_act_code(cv[2]))
            L["w_exp"], L["b_exp"] = (t.cpu() for t in _fold(cv[0], cv[1]))
            w, b = _fold(cv[3], cv[4])
            L["w_dw"], L["b_dw"] = w.view(cexp, k * k).t().cpu(), b.cpu()                       # [k*k][C]
            se = cv[5]
            if not isinstance(se, nn.Identity):
                L["se"] = dict(h=se.fc[0].out_features, w1=se.fc[0].weight.detach().double().cpu(), w2=se.fc[2].weight.detach().double().cpu(),
                               act1=_act_code(se.fc[1]), act2=_act_code(se.fc[3]))
            L["w_proj"], L["b_proj"] = (t.cpu() for t in _fold(cv[7], cv[8]))
            L["w_exp"], L["w_proj"] = L["w_exp"].view(cexp, cin), L["w_proj"].view(cout, cexp)
            self.blocks.append(L)
            cin = cout
        self.c_last, self.p_last = cin, _pad32(cin)
        lin = module.output_layer[1]
        self.n_out = lin.out_features
        self.w_head = torch.nn.functional.pad(lin.weight.detach().double(), (0, self.p_last - cin)).to(dev, bf16).contiguous()
        self.b_head = lin.bias.detach().to(dev, f32).contiguous()
        self._plans = {}

    # ------------------------------------------------------------------ layout, weights and buffers for one batch size / resolution
    def _plan(self, N, S):
        """Channel axis of every activation: padded to a multiple of 32 (one lafs_gemm_nt call per 1x1 convolution, K % 32 == 0) --
        except on the large maps (side >= PACK_MIN_SIDE), which are stored UNPADDED when their width is a multiple of 8: 16 channels
        at 56 x 56 padded to 32 doubled the bytes of the four largest tensors of the network.  A 1x1 convolution reads such a
        tensor g pixels per GEMM row ([M, c] viewed as [M / g, g c], g c % 32 == 0) against the block-diagonal weight diag(W, .., W):
        the same sums (the added products are exact zeros), no new kernel; depthwise / pooling / rescaling kernels take any width
        that is a multiple of 8."""
        key = (N, S)
        if key in self._plans:
            return self._plans[key]
        dev = self.device
        pad = lambda t, r, c: torch.nn.functional.pad(t, (0, c - t.shape[1], 0, r - t.shape[0]))

        def width(c, side, rows):                                  # stored width of a c-channel tensor on a side x side map
            g = _group(c) if c % 8 == 0 else None
            return c if (side >= PACK_MIN_SIDE and g is not None and rows % g == 0) else _pad32(c)

        def conv1x1(W, b, wi, wo, rows):                           # [cout, cin] fp64 -> device images for input / output widths wi / wo
            g = _group(wi)
            while 2 * g * max(wi, wo) <= 128 and rows % (2 * g) == 0:      # tiny convolutions (16 -> 16): fill the GEMM's 128-wide tile
                g *= 2
            cout, cin = W.shape
            Wd = torch.zeros(g * wo, g * wi, dtype=torch.float64)
            bd = torch.zeros(g * wo, dtype=torch.float64)
            for j in range(g):
                Wd[j * wo:j * wo + cout, j * wi:j * wi + cin] = W
                if b is not None:
                    bd[j * wo:j * wo + cout] = b
            return g, Wd.to(dev, bf16).contiguous(), (bd.to(dev, f32).contiguous() if b is not None else None)

        H = S // 2
        w0 = width(16, H, N * H * H)
        bufs = dict(x0=torch.empty(N * H * H, w0, device=dev, dtype=bf16), w0=w0, layers=[])
        wi = w0
        for L in self.blocks:
            Ho = (H + L["stride"] - 1) // L["stride"]
            we = width(L["cexp"], H, N * H * H)                    # the expanded tensor keeps its width through the depthwise convolution
            if (N * Ho * Ho) % (_group(we) or 1) != 0:
                we = _pad32(L["cexp"])
            wo = width(L["cout"], Ho, N * Ho * Ho)
            d = dict(H=H, Ho=Ho, we=we, wo=wo, e=torch.empty(N * H * H, we, device=dev, dtype=bf16),
                     d=torch.empty(N * Ho * Ho, we, device=dev, dtype=bf16), y=torch.empty(N * Ho * Ho, wo, device=dev, dtype=bf16))
            d["g_exp"], d["w_exp"], d["b_exp"] = conv1x1(L["w_exp"], L["b_exp"], wi, we, N * H * H)
            d["w_dw"] = pad(L["w_dw"], L["k"] * L["k"], we).to(dev, f32).contiguous()
            d["b_dw"] = torch.nn.functional.pad(L["b_dw"], (0, we - L["cexp"])).to(dev, f32).contiguous()
            d["g_proj"], d["w_proj"], d["b_proj"] = conv1x1(L["w_proj"], L["b_proj"], we, wo, N * Ho * Ho)
            if "se" in L:
                pe, ph = _pad32(L["cexp"]), _pad32(L["se"]["h"])
                d["pe"] = pe
                d["w1"] = pad(L["se"]["w1"], ph, pe).to(dev, bf16).contiguous()
                d["w2"] = pad(L["se"]["w2"], pe, ph).to(dev, bf16).contiguous()
                d["pool"] = torch.zeros(N, pe, device=dev, dtype=bf16)          # (channels past `we` stay zero)
                d["hid"] = torch.empty(N, ph, device=dev, dtype=bf16)
                d["gate"] = torch.empty(N, pe, device=dev, dtype=bf16)
            bufs["layers"].append(d)
            H, wi = Ho, wo
        bufs["Hlast"], bufs["wlast"] = H, wi
        bufs["feat"] = torch.zeros(N, self.p_last, device=dev, dtype=bf16)
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
        call("lafs_cnn_stem", _p(x), _p(self.stem_w), _p(self.stem_b), N, S, self.stem_act, _p(P["x0"]), P["w0"])
        cur = P["x0"]
        rows = lambda t, g: t.view(t.shape[0] // g, g * t.shape[1])            # g pixels per GEMM row
        for L, B in zip(self.blocks, P["layers"]):
            H, Ho, we = B["H"], B["Ho"], B["we"]
            g = B["g_exp"]
            ops.gemm_nt(rows(cur, g), B["w_exp"], _lib.EPI_BF16_ACT, bias=B["b_exp"], out=rows(B["e"], g), act=L["act"])
            se = L.get("se")
            call("lafs_cnn_dwconv", _p(B["e"]), _p(B["w_dw"]), _p(B["b_dw"]), N, H, H, we, L["k"], L["stride"],
                 -1 if se else L["act"], _p(B["d"]))
            if se:
                call("lafs_cnn_pool", _p(B["d"]), N, Ho * Ho, we, _p(B["pool"]), B["pe"])
                ops.gemm_nt(B["pool"], B["w1"], _lib.EPI_BF16_ACT, out=B["hid"], act=se["act1"])
                ops.gemm_nt(B["hid"], B["w2"], _lib.EPI_BF16_ACT, out=B["gate"], act=se["act2"])
                call("lafs_cnn_scale_act", _p(B["d"]), _p(B["gate"]), B["pe"], N, Ho * Ho, we, L["act"])
            g = B["g_proj"]
            ops.gemm_nt(rows(B["d"], g), B["w_proj"], _lib.EPI_BF16_ACT, bias=B["b_proj"], out=rows(B["y"], g),
                        aux=rows(cur, g) if L["residual"] else None, act=_lib.ACT_NONE)
            cur = B["y"]
        Hl = P["Hlast"]
        call("lafs_cnn_pool", _p(cur), N, Hl * Hl, P["wlast"], _p(P["feat"]), self.p_last)
        ops.gemm_nt(P["feat"], self.w_head, _lib.EPI_F32, bias=self.b_head, out=P["t"])
        return P["t"]

    __call__ = forward
