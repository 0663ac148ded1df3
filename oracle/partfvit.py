"""Oracle: Part-fViT backbone (lucidrains-style ViT used for fine-tune and as the reference's
actual LAFS student) -- fp32 CPU.  Test infrastructure only."""
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class PartFViTConfig:
    """ViT_face_landmark_patch8 geometry (face_pre_pro/ViT_face.py:560-617; train_largescale.py:542-561)."""
    patch_size: int = 8
    dim: int = 768
    depth: int = 12
    heads: int = 11
    dim_head: int = 64
    mlp_dim: int = 2048
    num_patches: int = 196
    ln_eps: float = 1e-5          # plain nn.LayerNorm(dim), ViT_face.py:117

    @property
    def inner(self):
        return self.heads * self.dim_head

    @property
    def scale(self):
        # quirk kept on purpose: the MODEL dim, not the head dim (ViT_face.py:145)
        return self.dim ** -0.5


def init_params(cfg: PartFViTConfig, gen: torch.Generator):
    D, I, H = cfg.dim, cfg.inner, cfg.mlp_dim
    pd = 3 * cfg.patch_size ** 2

    def u(*shape, fan_in):
        b = fan_in ** -0.5
        return (torch.rand(*shape, generator=gen) * 2 - 1) * b

    P = {
        "pos_embedding": torch.randn(1, cfg.num_patches + 1, D, generator=gen),
        "cls_token": torch.randn(1, 1, D, generator=gen),
        "patch_to_embedding.weight": u(D, pd, fan_in=pd), "patch_to_embedding.bias": u(D, fan_in=pd),
        "mlp_head.0.weight": torch.ones(D), "mlp_head.0.bias": torch.zeros(D),
    }
    for i in range(cfg.depth):
        a, f = f"transformer.layers.{i}.0.fn.", f"transformer.layers.{i}.1.fn."
        P[a + "norm.weight"] = torch.ones(D); P[a + "norm.bias"] = torch.zeros(D)
        P[a + "fn.to_qkv.weight"] = u(3 * I, D, fan_in=D)                 # no bias (ViT_face.py:147)
        P[a + "fn.to_out.0.weight"] = u(D, I, fan_in=I); P[a + "fn.to_out.0.bias"] = u(D, fan_in=I)
        P[f + "norm.weight"] = torch.ones(D); P[f + "norm.bias"] = torch.zeros(D)
        P[f + "fn.net.0.weight"] = u(H, D, fan_in=D); P[f + "fn.net.0.bias"] = u(H, fan_in=D)
        P[f + "fn.net.3.weight"] = u(D, H, fan_in=H); P[f + "fn.net.3.bias"] = u(D, fan_in=H)
    return P


def patches_from_image(x, p):
    """'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (ViT_face.py:760): channel is the FASTEST index of
    the patch vector -- a different flattening from the Conv2d patch-embed of the DINO ViT."""
    B, C, Hh, Ww = x.shape
    h, w = Hh // p, Ww // p
    x = x.view(B, C, h, p, w, p).permute(0, 2, 4, 3, 5, 1)
    return x.reshape(B, h * w, p * p * C)


def attention(P, pre, x, cfg):
    """ViT_face.Attention (ViT_face.py:140-182): bias-free qkv, chunk(3) then 'b n (h d) -> b h n d'."""
    B, N, _ = x.shape
    q, k, v = F.linear(x, P[pre + "to_qkv.weight"]).chunk(3, dim=-1)
    sp = lambda t: t.view(B, N, cfg.heads, cfg.dim_head).transpose(1, 2)
    q, k, v = sp(q), sp(k), sp(v)
    a = (q @ k.transpose(-1, -2) * cfg.scale).softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, cfg.inner)
    return F.linear(o, P[pre + "to_out.0.weight"], P[pre + "to_out.0.bias"])


def transformer(P, x, cfg, drop_scales=None, masks=None):
    """Transformer of Residual_droppath(PreNorm(.)) pairs (ViT_face.py:106-120, 184-213).
    drop_scales [depth, 2, B] as in oracle.vit.  Element dropout (nn.Dropout after to_out :150-153, after GELU and
    after the second Linear :126-133) is applied through explicit factor tensors: masks[(layer, site)] with site
    0 = to_out [B, N, D], 1 = GELU [B, N, mlp], 2 = fc2 [B, N, D], entries 0 or 1/(1-p); None = rate 0."""
    D = cfg.dim
    mk = lambda i, s, t: t if masks is None or (i, s) not in masks else t * masks[(i, s)].view(t.shape)
    for i in range(cfg.depth):
        a, f = f"transformer.layers.{i}.0.fn.", f"transformer.layers.{i}.1.fn."
        y = attention(P, a + "fn.", F.layer_norm(x, (D,), P[a + "norm.weight"], P[a + "norm.bias"], cfg.ln_eps), cfg)
        y = mk(i, 0, y)
        if drop_scales is not None:
            y = y * drop_scales[i, 0].view(-1, 1, 1)
        x = y + x
        h = F.layer_norm(x, (D,), P[f + "norm.weight"], P[f + "norm.bias"], cfg.ln_eps)
        h = mk(i, 1, F.gelu(F.linear(h, P[f + "fn.net.0.weight"], P[f + "fn.net.0.bias"])))
        h = mk(i, 2, F.linear(h, P[f + "fn.net.3.weight"], P[f + "fn.net.3.bias"]))
        if drop_scales is not None:
            h = h * drop_scales[i, 1].view(-1, 1, 1)
        x = h + x
    return x


def standard_coords(num_land, noise=None, shuffle_id=None, batch=1):
    """The `use_standcoord` landmarks of ViT_face_landmark_patch8.forward (ViT_face.py:717-742): the centres of the sqrt(n) x sqrt(n)
    grid of 8 x 8 cells (meshgrid in 'ij' order, so landmark k = (i, j) carries (x, y) = (8 i + 4, 8 j + 4)), repeated over the
    batch; `noise` [B, n, 2] = the reference's torch.randn(theta.shape) draw (x 3 px, Random_prob), `shuffle_id` [B, n] = its
    torch.randint draw (landmarks re-drawn with replacement, shuffle)."""
    r = int(round(num_land ** 0.5))
    rc = torch.arange(0, r, dtype=torch.float32) * 8 + 4
    cx, cy = torch.meshgrid(rc, rc, indexing="ij")
    theta = torch.stack((cx, cy), 2).view(1, -1, 2).repeat(batch, 1, 1)
    if noise is not None:
        theta = theta + noise * 3
    if shuffle_id is not None:
        theta = torch.gather(theta, 1, shuffle_id.view(batch, -1, 1).repeat(1, 1, 2))
    return theta


def forward_embedding(P, x, cfg, drop_scales=None, masks=None, return_tokens=False):
    """ViT_face_landmark_patch8.forward without the landmark branch: 4-D image or 3-D [B,n,192]
    patches -> emb [B, dim] (ViT_face.py:759-776).  masks: see transformer(); masks["emb"] [B, n+1, D] is the
    embedding dropout (:614, 768).  return_tokens: also the patch tokens behind the transformer, in front of the head's
    LayerNorm (`save_token`, :769-770)."""
    if x.dim() == 4:
        x = patches_from_image(x, cfg.patch_size)
    t = F.linear(x, P["patch_to_embedding.weight"], P["patch_to_embedding.bias"])
    B, n, _ = t.shape
    t = torch.cat((P["cls_token"].expand(B, -1, -1), t), dim=1)
    t = t + P["pos_embedding"][:, :n + 1]
    if masks is not None and "emb" in masks:
        t = t * masks["emb"].view(t.shape)
    t = transformer(P, t, cfg, drop_scales, masks)
    emb = F.layer_norm(t[:, 0], (cfg.dim,), P["mlp_head.0.weight"], P["mlp_head.0.bias"], cfg.ln_eps)
    return (emb, t[:, 1:]) if return_tokens else emb
