"""Drop-in surface of the reference's ``face_pre_pro/ViT_face.py`` for the symbols on the hot path
(SURVEY.md section 8b): ``CosFace``, ``ViT_face_landmark_patch8`` (Part-fViT) and
``extract_patches_pytorch_gridsample`` -- same constructor / forward signatures and state_dict keys, running on the
gfx950 HIP kernels.  Experiment variants the entry points never instantiate are not provided.
"""
import math

import torch
import torch.nn as nn

from .. import _lib, functional as Fn, ops
from ..ops import _p, call
from ..vision_transformer import attach_arena

f32, bf16 = torch.float32, torch.bfloat16


# ------------------------------------------------------------------------------------------------- landmark gather
class _PatchGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, imgs, landmarks, n):
        B, _, S, _ = imgs.shape
        r = int(math.isqrt(n))
        img = imgs.contiguous().float()
        th = landmarks[:, :n].contiguous().float()
        out = torch.empty(B, 3, 8 * r, 8 * r, device=imgs.device, dtype=f32)
        call("lafs_patch_gather_fwd", _p(img), _p(th), B, S, n, _p(out))
        ctx.save_for_backward(img, th)
        ctx.n, ctx.full_n = n, landmarks.shape[1]
        return out

    @staticmethod
    def backward(ctx, dout):
        img, th = ctx.saved_tensors
        B, _, S, _ = img.shape
        dth = torch.empty_like(th)
        dimg = torch.zeros_like(img) if ctx.needs_input_grad[0] else None
        call("lafs_patch_gather_bwd", _p(img), _p(th), _p(dout.contiguous().float()), B, S, ctx.n, _p(dth), _p(dimg))
        if ctx.full_n != ctx.n:
            full = torch.zeros(B, ctx.full_n, 2, device=th.device, dtype=f32)
            full[:, :ctx.n] = dth
            dth = full
        return dimg, dth, None


def extract_patches_pytorch_gridsample(imgs, landmarks, patch_shape=None, num_landm=49):
    """8x8 bilinear patches around ``landmarks`` (x, y in pixels) assembled into a sqrt(n) x sqrt(n) mosaic
    (reference face_pre_pro/ViT_face.py:1615-1656: n sequential grid_sample calls) as ONE kernel launch,
    differentiable with respect to the landmarks and the image.  Only 8x8 patches (the LAFS configuration)."""
    if patch_shape is not None and (int(patch_shape[0]) != 8 or int(patch_shape[1]) != 8):
        raise NotImplementedError("the HIP gather kernel is specialised for 8x8 patches")
    if imgs.shape[1] != 3:
        raise NotImplementedError("3-channel images only")
    return _PatchGather.apply(imgs, landmarks, int(num_landm))


# ------------------------------------------------------------------------------------------------- margin head
class _CosFaceFunction(torch.autograd.Function):
    """s * (cos(x, W) - m * y).  cos is an MFMA GEMM over row-normalised bf16 operands; the normalisations are HIP
    row kernels.  (The training engine fuses margin + softmax + CE instead: lafs_margin_softmax_ce.)"""

    @staticmethod
    def forward(ctx, x, weight, label, s, m):
        B, D = x.shape
        C = weight.shape[0]
        Cpad = (C + 127) // 128 * 128
        dev = x.device
        xn = torch.empty(B, D, device=dev, dtype=bf16); inv_x = torch.empty(B, device=dev, dtype=f32)
        x32 = x.contiguous().float()
        call("lafs_l2norm_fwd", _p(x32), D, _p(xn), D, _p(inv_x), B, D)
        wn = torch.empty(Cpad, D, device=dev, dtype=bf16); inv_w = torch.empty(C, device=dev, dtype=f32)
        ones = torch.ones(C, device=dev, dtype=f32)
        w32 = weight.contiguous().float()
        call("lafs_weightnorm_fwd", _p(w32), _p(ones), C, Cpad, D, _p(wn), None, Cpad, _p(inv_w))
        cos = ops.gemm_nt(xn, wn, _lib.EPI_F32, n_cols=Cpad)[:, :C]
        if label.dim() > 1:
            y = label.to(f32)
        else:
            y = torch.zeros(B, C, device=dev, dtype=f32).scatter_(1, label.view(-1, 1).long(), 1.0)
        ctx.save_for_backward(x32, w32, xn, wn, inv_x, inv_w, ones)
        ctx.s, ctx.C, ctx.Cpad = s, C, Cpad
        return s * (cos - m * y)

    @staticmethod
    def backward(ctx, dout):
        x32, w32, xn, wn, inv_x, inv_w, ones = ctx.saved_tensors
        B, D = x32.shape
        C, Cpad = ctx.C, ctx.Cpad
        dev = x32.device
        dcos = torch.zeros(B, Cpad, device=dev, dtype=bf16)
        dcos[:, :C] = (dout * ctx.s).to(bf16)
        # d(xn) = dcos @ wn  (contraction over classes -> split-K), d(wn) = dcos^T @ xn
        wn_t = wn.t().contiguous()
        dxn = ops.gemm_nt(dcos, wn_t, _lib.EPI_ATOMIC_F32, splits=max(1, min(64, Cpad // 1024)))
        dwn = torch.zeros(Cpad, D, device=dev, dtype=f32)
        ops.gemm_tn_acc(dcos, xn, dwn, splits=1)
        dx = torch.empty(B, D, device=dev, dtype=f32)
        call("lafs_l2norm_bwd", _p(x32), D, _p(dxn), D, _p(inv_x), _p(dx), D, B, D)
        dw = torch.empty(C, D, device=dev, dtype=f32)
        call("lafs_weightnorm_bwd", _p(dwn), _p(w32), _p(ones), _p(inv_w), C, D, _p(dw), None, 0)
        return dx, dw, None, None, None


class CosFace(nn.Module):
    """CosFace margin head (reference face_pre_pro/ViT_face.py:26-96): out = s * (cos(x, W) - m * label), where label is a
    class-index vector or a dense soft target (the mixup branch).  ``device_id`` must be None (the reference's
    single-process model-parallel branch is dead code: both entry points pass GPU_ID=None)."""

    def __init__(self, in_features, out_features, device_id, s=64.0, m=0.4):
        super().__init__()
        if device_id is not None:
            raise NotImplementedError("device_id model parallelism is not part of the hot path (always None in the reference)")
        if in_features % 64:
            raise NotImplementedError("in_features must be a multiple of 64 for the MFMA GEMM")
        self.in_features, self.out_features, self.device_id, self.s, self.m = in_features, out_features, device_id, s, m
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.xavier_uniform_(self.weight)

    def forward(self, input, label):
        return _CosFaceFunction.apply(input, self.weight, label, float(self.s), float(self.m))

    def __repr__(self):
        return f"{self.__class__.__name__}(in_features = {self.in_features}, out_features = {self.out_features}, s = {self.s}, m = {self.m})"


# ------------------------------------------------------------------------------------------------- Part-fViT
class _Holder(nn.Module):
    pass


class _PartFViTFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, hook, x, pos):
        save = any(ctx.needs_input_grad)
        n_img = x.shape[0]
        n = x.shape[1] if x.dim() == 3 else (x.shape[-1] // 8) ** 2
        side = 8 * int(math.isqrt(n))
        if (side // 8) ** 2 != n:
            raise _lib.LafsHipError("the packed engine needs a square number of patches per image")
        geom = Fn.geometry([(n_img, side)], x.device)
        drop = model._sample_drop_scales(geom) if model.training else None
        feat, st, _ = Fn.vit_forward(model._arena, model._spec, geom, [x.contiguous().float()], [pos.detach().contiguous()],
                                     drop, save=save, dropout=model._next_dropout())
        ctx.model, ctx.st = model, (st if save else None)
        ctx.x_dim = x.dim()
        model._last_tokens = _.view(n_img, n + 1, -1)[:, 1:] if getattr(model, "_want_tokens", False) else None
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        model, st = ctx.model, ctx.st
        if not ctx.needs_input_grad[2]:
            dpos = Fn.vit_backward(model._arena, model._spec, st, dfeat)
            return None, None, None, dpos[0]
        # landmark branch: the patch vectors themselves are differentiable (reference ViT_face.py:679-711)
        dpos, dx = Fn.vit_backward(model._arena, model._spec, st, dfeat, want_dx=True)
        dx = dx[0] if ctx.x_dim == 3 else Fn.unpatchify_grad(dx[0], model._spec.patch_order)
        return None, None, dx, dpos[0]


class ViT_face_landmark_patch8(nn.Module):
    """Part-fViT (reference face_pre_pro/ViT_face.py:560-795): pre-LN ViT with bias-free qkv, attention scale dim**-0.5,
    ``heads * 64`` inner width, DropPath 0.1 on every residual branch, LayerNorm head; 4-D images or 3-D [B, n, 192]
    patch vectors.  ``with_land=True`` adds the trainable MobileNetV3 landmark regressor (``stn`` + ``output_layer``, stock
    PyTorch-ROCm / MIOpen) whose landmarks drive the single-launch HIP patch gather; gradients flow back through the patch
    embedding and the gather into theta (reference :679-711).  Element dropout (``dropout`` after to_out / GELU / fc2,
    ``emb_dropout`` on the embedded tokens) is fused into the GEMM epilogues with counter-based masks (same distribution as
    nn.Dropout, different random stream)."""

    def __init__(self, *, loss_type, GPU_ID, num_class, image_size, patch_size, dim, depth, heads, mlp_dim, pool='cls',
                 num_patches=None, channels=3, dim_head=64, dropout=0., emb_dropout=0., fp16=True, with_land=False,
                 use_standcoord=False, Random_prob=False, shuffle=False, drop_path_rate=0.1):
        super().__init__()
        if patch_size != 8 or channels != 3 or dim_head != 64:
            raise NotImplementedError("HIP kernels are specialised for 3x8x8 patches and head_dim 64")
        if not (0.0 <= dropout < 1.0 and 0.0 <= emb_dropout < 1.0):
            raise ValueError("dropout rates must be in [0, 1)")
        if pool != 'cls':
            raise NotImplementedError("only cls pooling is on the hot path")
        if num_patches is None:
            num_patches = (image_size // patch_size) ** 2
        self.patch_size, self.fp16, self.num_patches = patch_size, fp16, num_patches
        self.row_num = int(math.sqrt(num_patches))
        self.with_land, self.pool, self.loss_type, self.GPU_ID = with_land, pool, loss_type, GPU_ID
        self.use_standcoord, self.Random_prob, self.shuffle = use_standcoord, Random_prob, shuffle      # reference :576, 717-742
        self._want_tokens, self._last_tokens = False, None
        if with_land:                                     # reference :577-602
            from .mobilenet import MobileNetV3_backbone
            self.stn = MobileNetV3_backbone(mode='large')
            self.output_layer = nn.Sequential(nn.Dropout(p=0.5), nn.Linear(160, self.row_num * self.row_num * 2))
        self.patch_shape = torch.tensor([patch_size, patch_size])
        self.dim, self.depth, self.heads, self.mlp_dim = dim, depth, heads, mlp_dim
        self.drop_path_rate = drop_path_rate
        # element dropout (after to_out, after GELU, after fc2, and on the embedded tokens): fused into the GEMM epilogues
        # with counter-based masks; `_drop_step` advances the seed every training forward
        self.dropout_rate, self.emb_dropout_rate, self._drop_seed0, self._drop_step = float(dropout), float(emb_dropout), 0x5EED, 0
        inner = heads * dim_head
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches + 1, dim))
        self.patch_to_embedding = nn.Linear(channels * patch_size ** 2, dim)
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.transformer = _Holder()
        layers = []
        for _ in range(depth):
            att = _Holder(); att.fn = _Holder(); att.fn.norm = nn.LayerNorm(dim); att.fn.fn = _Holder()
            att.fn.fn.to_qkv = nn.Linear(dim, inner * 3, bias=False)
            att.fn.fn.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))
            ff = _Holder(); ff.fn = _Holder(); ff.fn.norm = nn.LayerNorm(dim); ff.fn.fn = _Holder()
            ff.fn.fn.net = nn.Sequential(nn.Linear(dim, mlp_dim), nn.GELU(), nn.Dropout(dropout), nn.Linear(mlp_dim, dim), nn.Dropout(dropout))
            layers.append(nn.ModuleList([att, ff]))
        self.transformer.layers = nn.ModuleList(layers)
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim))
        if loss_type == 'None':
            pass
        elif loss_type == 'CosFace':
            self.loss = CosFace(in_features=dim, out_features=num_class, device_id=GPU_ID, m=0.4)
        else:
            raise NotImplementedError(f"loss_type {loss_type!r}: only 'CosFace' and 'None' exist in the reference")
        self.theta = 0
        self._arena, self._spec, self._hook = None, None, None

    def _bind_arena(self, arena, prefix):
        names = []
        for i in range(self.depth):
            a, f = f"{prefix}transformer.layers.{i}.0.fn.", f"{prefix}transformer.layers.{i}.1.fn."
            names.append(dict(ln1_g=a + "norm.weight", ln1_b=a + "norm.bias", w_qkv=a + "fn.to_qkv.weight", b_qkv=None,
                              w_proj=a + "fn.to_out.0.weight", b_proj=a + "fn.to_out.0.bias",
                              ln2_g=f + "norm.weight", ln2_b=f + "norm.bias",
                              w_fc1=f + "fn.net.0.weight", b_fc1=f + "fn.net.0.bias",
                              w_fc2=f + "fn.net.3.weight", b_fc2=f + "fn.net.3.bias"))
        trunk = Fn.TrunkSpec(dim=self.dim, heads=self.heads, mlp=self.mlp_dim, depth=self.depth, ln_eps=1e-5,
                             attn_scale=self.dim ** -0.5, block_names=names)      # scale quirk: model dim (ref :145)
        object.__setattr__(self, "_arena", arena)
        self._spec = Fn.ViTSpec(trunk=trunk, prefix=prefix, patch_order=_lib.PATCH_ORDER_HWC,
                                w_patch="patch_to_embedding.weight", b_patch="patch_to_embedding.bias", cls="cls_token",
                                final_g="mlp_head.0.weight", final_b="mlp_head.0.bias", pos="pos_embedding")
        self._hook = torch.zeros(1, device=arena.device, requires_grad=True)

    def _sample_drop_scales(self, geom):
        """Residual_droppath: the same rate on both branches of every layer (reference :106-112)."""
        if not self.drop_path_rate:
            return None
        keep = 1.0 - self.drop_path_rate
        u = torch.rand(self.depth, 2, geom.n_seq, device=self._arena.device)
        return (torch.floor(keep + u) / keep).contiguous()

    def _next_dropout(self):
        """(p_trunk, p_embedding, seed) for this forward, or None in eval mode / at rate 0."""
        if not self.training or (self.dropout_rate == 0.0 and self.emb_dropout_rate == 0.0):
            return None
        self._drop_step += 1
        return (self.dropout_rate, self.emb_dropout_rate, (self._drop_seed0 + 7919 * self._drop_step) & 0x3FFFFFFF)

    def forward_embedding(self, x):
        if self._arena is None:
            attach_arena(self)
        self._arena.ensure_fresh()
        n = x.shape[1] if x.dim() == 3 else (x.shape[-1] // 8) ** 2
        pos = self.pos_embedding[0, :n + 1]
        return _PartFViTFunction.apply(self, self._hook, x, pos)

    def landmarks(self, x):
        """[B,3,112,112] -> theta [B, r*r, 2] in pixels: CNN -> mean pool -> Dropout(0.5)+Linear(160, 2*r*r) -> per-sample
        min-max to [0, 111] (reference :680-706).  Differentiable (torch autograd over MIOpen convolutions)."""
        t = self.output_layer(self.stn(x).mean(dim=(-2, -1)))
        tmax, tmin = t.max(dim=1, keepdim=True)[0], t.min(dim=1, keepdim=True)[0]
        return ((t - tmin) / (tmax - tmin) * 111).view(-1, self.row_num * self.row_num, 2)

    def forward(self, x, label=None, mask=None, visualize=False, save_token=False, opt=None, keep_num=None, glo_diff=False):
        """(reference :659-795)  `save_token`: also the patch tokens behind the transformer, in front of the head's LayerNorm, detached
        (an analysis dump in the reference: returns (emb, tokens, theta)); `use_standcoord` (constructor): the patches are gathered at
        the centres of the regular 8 x 8 grid -- jittered by N(0, 3^2) px with `Random_prob`, re-drawn with replacement with `shuffle`,
        both from the CPU generator in the reference's order -- and the mosaic is transposed (:717-742).  Attention masks are not
        supported (never passed by either entry point)."""
        if mask is not None:
            raise NotImplementedError("attention masks are not on the training hot path (never passed by either entry point)")
        if self._arena is None:
            attach_arena(self)
        theta = self.theta
        if self.with_land and x.dim() == 4:
            num_land = keep_num if keep_num is not None else self.row_num * self.row_num
            theta = self.landmarks(x)
            self.theta = theta
            x = extract_patches_pytorch_gridsample(x, theta[:, :num_land], patch_shape=self.patch_shape, num_landm=num_land)
        if self.use_standcoord and x.dim() == 4:
            b, side = x.shape[0], x.shape[-1]
            num_land = (side // self.patch_size) ** 2
            rc = torch.arange(0, int(math.isqrt(num_land)), dtype=torch.float32) * 8 + 4
            cx, cy = torch.meshgrid(rc, rc, indexing="ij")
            theta = torch.stack((cx, cy), 2).view(1, -1, 2).repeat(b, 1, 1)
            if self.Random_prob:
                theta = theta + torch.randn(theta.shape) * 3
            if self.shuffle:
                ids = torch.randint(0, theta.shape[1], (b, theta.shape[1], 1))
                theta = torch.gather(theta, 1, ids.repeat(1, 1, 2))
            theta = theta.to(x.device)
            x = extract_patches_pytorch_gridsample(x, theta[:, :num_land], patch_shape=self.patch_shape, num_landm=num_land)
            x = x.permute(0, 1, 3, 2).contiguous()
        self._want_tokens = bool(save_token)
        emb = self.forward_embedding(x)
        self._want_tokens = False
        if save_token:
            return emb, self._last_tokens, self.theta
        if label is not None:
            return self.loss(emb, label), self.theta
        return (emb, theta) if visualize else emb


# ------------------------------------------------------------------------------------------------- landmark CNN wrapper
class face_landmark_4simmin_glo_loc(nn.Module):
    """Frozen landmark regressor of the LAFS step (reference face_pre_pro/ViT_face.py:1218-1409): MobileNetV3 trunk ->
    mean pool -> Linear(160, 2*r*r) -> per-sample min-max to [0, 111] px -> optional N(0, 5^2) px jitter -> optional random
    choice of 36 landmarks (with replacement) -> 8x8 bilinear patch gather from ``x_Aug`` into a mosaic image.
    The CNN runs on stock PyTorch-ROCm; the gather is the single-launch HIP kernel.  Random draws are made on the CPU
    generator in the reference's order (randn for the jitter, then randint for the selection) and moved to the device."""

    def __init__(self, *, loss_type, GPU_ID, num_class, image_size, patch_size, dim, depth, heads, mlp_dim, pool='cls',
                 num_patches=None, channels=3, dim_head=64, dropout=0., emb_dropout=0., fp16=True):
        super().__init__()
        from .mobilenet import MobileNetV3_backbone
        if num_patches is None:
            num_patches = (image_size // patch_size) ** 2
        if patch_size != 8:
            raise NotImplementedError("the HIP gather kernel is specialised for 8x8 patches")
        self.patch_size, self.fp16, self.num_patches, self.dim = patch_size, fp16, num_patches, dim
        self.row_num = int(math.sqrt(num_patches))
        self.stn = MobileNetV3_backbone(mode='large')
        self.output_layer = nn.Sequential(nn.Dropout(p=0.5), nn.Linear(160, self.row_num * self.row_num * 2))
        self.global_token = nn.Sequential(nn.Dropout(p=0.5), nn.Linear(160, dim))
        self.patch_shape = torch.tensor([patch_size, patch_size])
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches + 1, dim))
        self.patch_to_embedding = nn.Linear(channels * patch_size ** 2, dim)
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.mask_token = nn.Parameter(torch.zeros(1, 1, dim))
        nn.init.trunc_normal_(self.mask_token, std=.02, a=-.02, b=.02)
        self.loss_type, self.GPU_ID, self.num_features, self.in_chans = loss_type, GPU_ID, dim, channels
        self.theta = 0

    def landmarks(self, x):
        """[B,3,S,S] -> theta [B, r*r, 2] in pixels (x, y), min-max scaled per sample (reference :1338-1355)."""
        t = self.output_layer(self.stn(x).mean(dim=(-2, -1)))
        tmax, tmin = t.max(dim=1, keepdim=True)[0], t.min(dim=1, keepdim=True)[0]
        return ((t - tmin) / (tmax - tmin) * 111).view(-1, self.row_num * self.row_num, 2)

    def forward(self, x, x_Aug=None, keep_num=None, patch_shape=None, Random_prob=False, return_prob=False, ran_sample=False,
                random_coor=False, return_land=False):
        if random_coor:
            num_land = 25 if ran_sample else self.row_num * self.row_num
            theta = (torch.rand(x.shape[0], num_land, 2) * 111.0).to(x.device)
        else:
            S = x.shape[-2]
            num_land = self.num_patches if (S == 112 and self.num_patches in (144, 196)) else (S // self.patch_size) ** 2
            theta = self.landmarks(x)
            if Random_prob:
                theta = theta + (torch.randn(theta.shape) * 5).to(theta.device)
                if not return_prob:
                    b, c, _ = theta.shape
                    num_land = 36 if ran_sample else self.row_num * self.row_num
                    idx = torch.randint(0, c, (b, num_land, 1)).to(theta.device).repeat(1, 1, 2)
                    theta = torch.gather(theta, 1, idx)
            self.theta = theta
        if return_land:
            return theta, x
        src = x if x_Aug is None else x_Aug
        return theta, extract_patches_pytorch_gridsample(src, theta[:, :num_land], patch_shape=self.patch_shape, num_landm=num_land)
