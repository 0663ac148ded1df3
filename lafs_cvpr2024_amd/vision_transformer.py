"""Drop-in module surface of the reference's ``vision_transformer.py`` on top of the gfx950 HIP kernels.

Same class names, constructor/forward signatures and ``state_dict`` keys as the reference
(vision_transformer.py:134-159 VisionTransformer, :237-262 vit_tiny/small/base, :265-301 DINOHead), but the
sub-modules here only HOLD parameters: ``forward`` packs all crops into one token batch and runs the C engine
(``lafs_trunk_forward`` / ``lafs_trunk_backward``).  There is no ATen math on the path except the bicubic resampling
of the position table (a 785x384 tensor), which stays in torch so its gradient reaches ``pos_embed``.
"""
import math
from functools import partial

import torch
import torch.nn as nn

from . import _lib, functional as Fn
from .arena import ParamArena

__all__ = ["VisionTransformer", "DINOHead", "vit_tiny", "vit_small", "vit_base", "attach_arena"]


def _is_matrix_for_dgrad(name, p):
    """Weights that need a transposed bf16 shadow (their dgrad runs as an NT GEMM on W^T)."""
    return p.dim() == 2 and any(k in name for k in ("attn.", "mlp.", "transformer.layers."))


def attach_arena(module, device=None):
    """Move every parameter of ``module`` (e.g. a MultiCropWrapper) into one ParamArena and tell the kernel-driving
    sub-modules where they live.  Idempotent."""
    arena = getattr(module, "_lafs_arena", None)
    if arena is not None:
        return arena
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    arena = ParamArena(module, device, with_grad=True, transposed=_is_matrix_for_dgrad)
    object.__setattr__(module, "_lafs_arena", arena)
    for mod_name, m in module.named_modules():
        if hasattr(m, "_bind_arena"):
            m._bind_arena(arena, mod_name + "." if mod_name else "")
    for b in module.buffers():
        b.data = b.data.to(device)
    return arena


class _ViTFunction(torch.autograd.Function):
    """Whole packed ViT pass as one autograd node.  Parameter gradients are accumulated by the kernels straight into
    the arena (``p.grad`` are views of it); only the resampled position tables are differentiable inputs."""

    @staticmethod
    def forward(ctx, vit, n_groups, hook, *tensors):
        imgs, pos = tensors[:n_groups], tensors[n_groups:]
        geom = Fn.geometry([(im.shape[0], im.shape[-1]) for im in imgs], imgs[0].device)
        save = any(ctx.needs_input_grad)
        drop = vit._sample_drop_scales(geom) if vit.training else None
        feat, st, _ = Fn.vit_forward(vit._arena, vit._spec, geom, [im.contiguous().float() for im in imgs],
                                     [p.detach().contiguous() for p in pos], drop, save=save)
        ctx.vit, ctx.st, ctx.n_groups = vit, (st if save else None), n_groups
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        vit, st = ctx.vit, ctx.st
        dpos = Fn.vit_backward(vit._arena, vit._spec, st, dfeat)
        return (None, None, None) + (None,) * ctx.n_groups + tuple(dpos)


class VisionTransformer(nn.Module):
    """Vision Transformer (reference vision_transformer.py:134-215).  patch_size 8 and head_dim 64 only."""

    def __init__(self, img_size=[224], patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=nn.LayerNorm, **kwargs):
        super().__init__()
        if patch_size != 8 or in_chans != 3:
            raise NotImplementedError("the HIP patch-embed kernel is specialised for 3x8x8 patches (LAFS uses --patch_size 8)")
        if embed_dim % num_heads or embed_dim // num_heads != 64:
            raise NotImplementedError("the HIP attention kernels are specialised for head_dim 64")
        if drop_rate or attn_drop_rate:
            raise NotImplementedError("dropout inside the blocks is not on the LAFS path (rates are 0 in the reference)")
        self.num_features = self.embed_dim = embed_dim
        self.depth, self.num_heads = depth, num_heads
        self.qk_scale = qk_scale or (embed_dim // num_heads) ** -0.5
        self.drop_path_rates = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        hidden = int(embed_dim * mlp_ratio)

        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.patch_embed.patch_size = patch_size
        self.patch_embed.num_patches = (img_size[0] // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        blocks = []
        for _ in range(depth):
            blk = nn.Module()
            blk.norm1 = norm_layer(embed_dim)
            blk.attn = nn.Module()
            blk.attn.qkv = nn.Linear(embed_dim, embed_dim * 3, bias=qkv_bias)
            blk.attn.proj = nn.Linear(embed_dim, embed_dim)
            blk.norm2 = norm_layer(embed_dim)
            blk.mlp = nn.Module()
            blk.mlp.fc1 = nn.Linear(embed_dim, hidden)
            blk.mlp.fc2 = nn.Linear(hidden, embed_dim)
            blocks.append(blk)
        self.blocks = nn.ModuleList(blocks)
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self.fc = nn.Identity()

        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.zeros_(m.bias)
                nn.init.ones_(m.weight)
        self._arena, self._spec = None, None
        self._hook = None

    # ------------------------------------------------------------------ arena binding
    def _bind_arena(self, arena, prefix):
        names = []
        for i in range(self.depth):
            b = f"{prefix}blocks.{i}."
            names.append(dict(ln1_g=b + "norm1.weight", ln1_b=b + "norm1.bias", w_qkv=b + "attn.qkv.weight",
                              b_qkv=(b + "attn.qkv.bias") if self.blocks[i].attn.qkv.bias is not None else None,
                              w_proj=b + "attn.proj.weight", b_proj=b + "attn.proj.bias",
                              ln2_g=b + "norm2.weight", ln2_b=b + "norm2.bias",
                              w_fc1=b + "mlp.fc1.weight", b_fc1=b + "mlp.fc1.bias",
                              w_fc2=b + "mlp.fc2.weight", b_fc2=b + "mlp.fc2.bias"))
        trunk = Fn.TrunkSpec(dim=self.embed_dim, heads=self.num_heads, mlp=self.blocks[0].mlp.fc1.out_features,
                             depth=self.depth, ln_eps=self.norm.eps, attn_scale=self.qk_scale, block_names=names)
        object.__setattr__(self, "_arena", arena)
        self._spec = Fn.ViTSpec(trunk=trunk, prefix=prefix)
        self._hook = torch.zeros(1, device=arena.device, requires_grad=True)
        self._keep_prob = 1.0 - torch.tensor(self.drop_path_rates, device=arena.device, dtype=torch.float32).view(-1, 1, 1)

    def _ensure_arena(self):
        if self._arena is None:
            attach_arena(self)
        self._arena.ensure_fresh()

    def _sample_drop_scales(self, geom):
        """Per-sample stochastic-depth scales 0 or 1/keep for both residual branches of every block
        (reference drop_path, vision_transformer.py:27-35): [depth, 2, n_seq] f32, or None when all rates are 0."""
        if not any(self.drop_path_rates):
            return None
        keep = self._keep_prob                       # device tensor made at bind time (no H2D inside graph capture)
        u = torch.rand(self.depth, 2, geom.n_seq, device=self._arena.device)
        return (torch.floor(keep + u) / keep).contiguous()

    # ------------------------------------------------------------------ reference API
    def interpolate_pos_encoding(self, x, w, h):
        """Bicubic resampling of the stored table to the crop's patch grid (reference :174-194), in torch."""
        npatch = x.shape[1] - 1
        N = self.pos_embed.shape[1] - 1
        if npatch == N and w == h:
            return self.pos_embed
        dim = self.pos_embed.shape[-1]
        p = self.patch_embed.patch_size
        g = int(math.sqrt(N))
        w0, h0 = w // p + 0.1, h // p + 0.1
        grid = self.pos_embed[:, 1:].reshape(1, g, g, dim).permute(0, 3, 1, 2)
        grid = nn.functional.interpolate(grid, scale_factor=(w0 / g, h0 / g), mode="bicubic")
        assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
        return torch.cat((self.pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)

    def forward_groups(self, groups):
        """Several crop batches of different resolutions in ONE packed pass -> features [sum B, D]."""
        self._ensure_arena()
        pos = []
        for im in groups:
            if im.dim() != 4:
                raise _lib.LafsHipError("VisionTransformer expects NCHW crops")
            n = (im.shape[-1] // 8) ** 2
            pos.append(self.interpolate_pos_encoding(torch.empty(1, n + 1, 0), im.shape[-2], im.shape[-1])[0])
        return _ViTFunction.apply(self, len(groups), self._hook, *groups, *pos)

    def forward(self, x):
        return self.forward_groups([x])


def vit_tiny(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_small(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_base(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


# ------------------------------------------------------------------------------------------------- DINO head
class _HeadFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head, hook, x):
        save = any(ctx.needs_input_grad)
        logits, st = Fn.head_forward(head._arena, head._prefix, x.float(), head.out_dim, save=save)
        ctx.head, ctx.st = head, (st if save else None)
        return logits[:, :head.out_dim]

    @staticmethod
    def backward(ctx, dlogits):
        head, st = ctx.head, ctx.st
        n = dlogits.shape[0]
        dl = torch.zeros(n, st.Kpad, device=dlogits.device, dtype=torch.bfloat16)
        src = dlogits.float()
        if src.stride(-1) != 1 or src.stride(0) % 4:
            src = src.contiguous()
        if st.K % 4 == 0 and src.stride(0) % 4 == 0:
            Fn.ops.scale_cast_bf16(src, out=dl[:, :st.K])
        else:                                       # odd class counts: plain dtype cast of the incoming gradient
            dl[:, :st.K].copy_(src)
        dx = Fn.head_backward(head._arena, head._prefix, st, dl, train_g=head.last_layer.weight_g.requires_grad)
        return None, None, dx


class _WeightNormLinear(nn.Module):
    """Parameter holder with the key layout of nn.utils.weight_norm(nn.Linear(in, out, bias=False)):
    ``weight_g`` [out, 1] and ``weight_v`` [out, in]."""

    def __init__(self, in_features, out_features):
        super().__init__()
        lin = nn.Linear(in_features, out_features, bias=False)
        self.weight_g = nn.Parameter(lin.weight.detach().norm(dim=1, keepdim=True))
        self.weight_v = nn.Parameter(lin.weight.detach().clone())


class DINOHead(nn.Module):
    """DINO projection head (reference vision_transformer.py:265-301): 3-layer GELU MLP, L2 normalisation and a
    weight-normalised bias-free last layer.  nlayers=3 / use_bn=False only (the LAFS configuration)."""

    def __init__(self, in_dim, out_dim, use_bn=False, norm_last_layer=True, nlayers=3, hidden_dim=2048, bottleneck_dim=256):
        super().__init__()
        if use_bn or nlayers != 3:
            raise NotImplementedError("the HIP DINO head implements the LAFS configuration: nlayers=3, use_bn=False")
        self.in_dim, self.out_dim = in_dim, out_dim
        self.mlp = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, hidden_dim), nn.GELU(),
                                 nn.Linear(hidden_dim, bottleneck_dim))
        for m in self.mlp:
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)
        self.last_layer = _WeightNormLinear(bottleneck_dim, out_dim)
        self.last_layer.weight_g.data.fill_(1)
        if norm_last_layer:
            self.last_layer.weight_g.requires_grad = False
        self._arena, self._prefix, self._hook = None, "", None

    def _bind_arena(self, arena, prefix):
        object.__setattr__(self, "_arena", arena)
        self._prefix = prefix
        self._hook = torch.zeros(1, device=arena.device, requires_grad=True)

    def forward(self, x):
        if self._arena is None:
            attach_arena(self)
        self._arena.ensure_fresh()
        return _HeadFunction.apply(self, self._hook, x)
