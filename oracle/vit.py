"""Oracle: DINO/timm VisionTransformer + DINOHead + multi-crop grouping (fp32 CPU).

Functional restatement over a ``{state_dict key: tensor}`` mapping so that the
same dict can be loaded into the product modules.  Test infrastructure only.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class ViTConfig:
    """Geometry of one VisionTransformer (vision_transformer.py:134-159, 237-262)."""
    patch_size: int = 8
    embed_dim: int = 384
    depth: int = 12
    num_heads: int = 6
    mlp_ratio: float = 4.0
    ln_eps: float = 1e-6          # partial(nn.LayerNorm, eps=1e-6), vision_transformer.py:240
    img_size: int = 224           # size of the stored pos_embed grid (lafs_train.py:201-205)
    qkv_bias: bool = True

    @property
    def head_dim(self):
        return self.embed_dim // self.num_heads

    @property
    def grid(self):
        return self.img_size // self.patch_size


VIT_TINY = dict(embed_dim=192, depth=12, num_heads=3)
VIT_SMALL = dict(embed_dim=384, depth=12, num_heads=6)
VIT_BASE = dict(embed_dim=768, depth=12, num_heads=12)


def init_vit_params(cfg: ViTConfig, gen: torch.Generator):
    """Random parameters with the reference's initialisers (vision_transformer.py:161-172)."""
    D, H = cfg.embed_dim, int(cfg.embed_dim * cfg.mlp_ratio)
    p = cfg.patch_size

    def tn(*shape):
        t = torch.empty(*shape)
        return torch.nn.init.trunc_normal_(t, std=.02, generator=gen)

    P = {
        "cls_token": tn(1, 1, D),
        "pos_embed": tn(1, cfg.grid * cfg.grid + 1, D),
        # Conv2d keeps its default (kaiming-uniform) init in the reference: _init_weights
        # only touches Linear/LayerNorm.  Any distribution is fine for parity tests.
        "patch_embed.proj.weight": (torch.rand(D, 3, p, p, generator=gen) - .5) * (2 / math.sqrt(3 * p * p)),
        "patch_embed.proj.bias": (torch.rand(D, generator=gen) - .5) * (2 / math.sqrt(3 * p * p)),
        "norm.weight": torch.ones(D), "norm.bias": torch.zeros(D),
    }
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        P[b + "norm1.weight"] = torch.ones(D); P[b + "norm1.bias"] = torch.zeros(D)
        P[b + "attn.qkv.weight"] = tn(3 * D, D)
        if cfg.qkv_bias:
            P[b + "attn.qkv.bias"] = torch.zeros(3 * D)
        P[b + "attn.proj.weight"] = tn(D, D); P[b + "attn.proj.bias"] = torch.zeros(D)
        P[b + "norm2.weight"] = torch.ones(D); P[b + "norm2.bias"] = torch.zeros(D)
        P[b + "mlp.fc1.weight"] = tn(H, D); P[b + "mlp.fc1.bias"] = torch.zeros(H)
        P[b + "mlp.fc2.weight"] = tn(D, H); P[b + "mlp.fc2.bias"] = torch.zeros(D)
    return P


def interp_pos_embed(pos_embed, n_patch_tokens, w, h, patch_size):
    """Bicubic resample of the stored position table (vision_transformer.py:174-194).

    ``pos_embed`` is [1, 1+g*g, D]; the patch part is resized from g x g to
    (w // p) x (h // p) with ``scale_factor = (w//p + 0.1) / g`` -- the 0.1 is the
    reference's guard against float truncation (:184-186).
    """
    N = pos_embed.shape[1] - 1
    if n_patch_tokens == N and w == h:
        return pos_embed
    D = pos_embed.shape[-1]
    g = int(math.sqrt(N))
    w0, h0 = w // patch_size + 0.1, h // patch_size + 0.1
    grid = pos_embed[:, 1:].reshape(1, g, g, D).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=(w0 / g, h0 / g), mode="bicubic")
    assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, D)
    return torch.cat((pos_embed[:, :1], grid), dim=1)


def patch_embed(P, x, patch_size):
    """Conv2d(3, D, k=p, s=p) -> flatten -> transpose (vision_transformer.py:126-131)."""
    y = F.conv2d(x, P["patch_embed.proj.weight"], P["patch_embed.proj.bias"], stride=patch_size)
    return y.flatten(2).transpose(1, 2)


def attention(P, prefix, x, num_heads):
    """Multi-head self-attention, scale = head_dim**-0.5 (vision_transformer.py:68-92)."""
    B, N, C = x.shape
    d = C // num_heads
    qkv = F.linear(x, P[prefix + "qkv.weight"], P.get(prefix + "qkv.bias"))
    qkv = qkv.reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)) * (d ** -0.5)
    a = a.softmax(dim=-1)
    y = (a @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(y, P[prefix + "proj.weight"], P[prefix + "proj.bias"])


def block(P, i, x, cfg, scale_attn=None, scale_mlp=None):
    """Pre-LN block with per-sample stochastic-depth scales (vision_transformer.py:107-113, 27-35).

    ``scale_*`` is None (no drop-path) or a [B] tensor holding 0 or 1/keep_prob.
    """
    b = f"blocks.{i}."
    D = cfg.embed_dim
    y = attention(P, b + "attn.", F.layer_norm(x, (D,), P[b + "norm1.weight"], P[b + "norm1.bias"], cfg.ln_eps),
                  cfg.num_heads)
    if scale_attn is not None:
        y = y * scale_attn.view(-1, 1, 1)
    x = x + y
    h = F.layer_norm(x, (D,), P[b + "norm2.weight"], P[b + "norm2.bias"], cfg.ln_eps)
    h = F.linear(h, P[b + "mlp.fc1.weight"], P[b + "mlp.fc1.bias"])
    h = F.gelu(h)                                             # nn.GELU default = exact erf
    h = F.linear(h, P[b + "mlp.fc2.weight"], P[b + "mlp.fc2.bias"])
    if scale_mlp is not None:
        h = h * scale_mlp.view(-1, 1, 1)
    return x + h


def vit_forward(P, x, cfg: ViTConfig, drop_scales=None):
    """VisionTransformer.forward: NCHW crops -> cls feature [B, D] (vision_transformer.py:196-215).

    ``drop_scales``: None or tensor [depth, 2, B] of per-sample residual-branch scales.
    """
    B, _, w, h = x.shape
    t = patch_embed(P, x, cfg.patch_size)
    t = torch.cat((P["cls_token"].expand(B, -1, -1), t), dim=1)
    t = t + interp_pos_embed(P["pos_embed"], t.shape[1] - 1, w, h, cfg.patch_size)
    for i in range(cfg.depth):
        sa = sm = None
        if drop_scales is not None:
            sa, sm = drop_scales[i, 0], drop_scales[i, 1]
        t = block(P, i, t, cfg, sa, sm)
    t = F.layer_norm(t, (cfg.embed_dim,), P["norm.weight"], P["norm.bias"], cfg.ln_eps)
    return t[:, 0]


# ----------------------------------------------------------------------------- DINO head
def init_head_params(in_dim, out_dim, gen, hidden_dim=2048, bottleneck_dim=256):
    """DINOHead parameters, nlayers=3, use_bn=False (vision_transformer.py:265-293)."""
    def tn(*shape):
        return torch.nn.init.trunc_normal_(torch.empty(*shape), std=.02, generator=gen)
    bound = 1 / math.sqrt(bottleneck_dim)
    return {
        "mlp.0.weight": tn(hidden_dim, in_dim), "mlp.0.bias": torch.zeros(hidden_dim),
        "mlp.2.weight": tn(hidden_dim, hidden_dim), "mlp.2.bias": torch.zeros(hidden_dim),
        "mlp.4.weight": tn(bottleneck_dim, hidden_dim), "mlp.4.bias": torch.zeros(bottleneck_dim),
        # last_layer is created AFTER self.apply(_init_weights) so it keeps nn.Linear's default init
        "last_layer.weight_g": torch.ones(out_dim, 1),
        "last_layer.weight_v": (torch.rand(out_dim, bottleneck_dim, generator=gen) * 2 - 1) * bound,
    }


def dino_head_forward(P, x, prefix=""):
    """3-layer GELU MLP -> L2 normalise -> weight-normed linear (vision_transformer.py:295-301, 284)."""
    h = F.gelu(F.linear(x, P[prefix + "mlp.0.weight"], P[prefix + "mlp.0.bias"]))
    h = F.gelu(F.linear(h, P[prefix + "mlp.2.weight"], P[prefix + "mlp.2.bias"]))
    h = F.linear(h, P[prefix + "mlp.4.weight"], P[prefix + "mlp.4.bias"])
    h = F.normalize(h, dim=-1, p=2)
    v, g = P[prefix + "last_layer.weight_v"], P[prefix + "last_layer.weight_g"]
    w = v * (g / v.norm(dim=1, keepdim=True))                 # old-style nn.utils.weight_norm, dim=0
    return F.linear(h, w)


# ----------------------------------------------------------------------------- multi-crop
def crop_groups(crops):
    """End indices of runs of consecutive equal-resolution crops (utils.py:618-629).

    4-D crops are keyed by their last dim, 3-D patch tensors by the token count.
    """
    key = (lambda t: t.shape[-1]) if crops[0].dim() >= 4 else (lambda t: t.shape[-2])
    ends, prev = [], None
    for i, c in enumerate(crops):
        k = key(c)
        if prev is not None and k != prev:
            ends.append(i)
        prev = k
    ends.append(len(crops))
    return ends


def multicrop_forward(Pb, Ph, crops, cfg: ViTConfig, drop_scales=None):
    """MultiCropWrapper.forward: one backbone pass per resolution group, one head pass
    (utils.py:610-659).  ``drop_scales`` is a list with one entry per group (or None)."""
    if not isinstance(crops, (list, tuple)):
        crops = [crops]
    feats, start = [], 0
    for gi, end in enumerate(crop_groups(crops)):
        ds = None if drop_scales is None else drop_scales[gi]
        feats.append(vit_forward(Pb, torch.cat(crops[start:end]), cfg, ds))
        start = end
    return dino_head_forward(Ph, torch.cat(feats))
