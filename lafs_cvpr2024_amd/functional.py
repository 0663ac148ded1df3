"""Kernel-level forward/backward passes of the hot-path models over ParamArena buffers (no autograd, no ATen math).

These functions are the single implementation used by the drop-in nn.Modules (through torch.autograd.Function
wrappers) and by the fused, hipGraph-captured training engine.
"""
import os
import ctypes as C
import math
from dataclasses import dataclass, field

import torch

from . import _lib
from . import ops
from .ops import _p, call

bf16 = torch.bfloat16
f32 = torch.float32


# ----------------------------------------------------------------------------------------------- packed geometry
@dataclass
class PackedGeometry:
    """Token packing of several equal-resolution crop groups into one [n_tok, D] batch."""
    groups: tuple                       # ((n_images, side), ...)
    patch: int = 8
    device: object = None
    n_seq: int = 0
    n_tok: int = 0
    max_len: int = 0
    tok_start: list = field(default_factory=list)     # first token row of each group
    seq_start: list = field(default_factory=list)     # first sequence index of each group
    cu_seqlens: torch.Tensor = None
    row2seq: torch.Tensor = None

    def __post_init__(self):
        cu, r2s, tok, seq = [0], [], 0, 0
        for n_img, side in self.groups:
            n = (side // self.patch) ** 2 + 1
            self.tok_start.append(tok)
            self.seq_start.append(seq)
            for _ in range(n_img):
                cu.append(cu[-1] + n)
            r2s.append(torch.arange(seq, seq + n_img, dtype=torch.int32).repeat_interleave(n))
            tok += n_img * n
            seq += n_img
            self.max_len = max(self.max_len, n)
        self.n_seq, self.n_tok = seq, tok
        self.cu_seqlens = torch.tensor(cu, dtype=torch.int32, device=self.device)
        self.row2seq = torch.cat(r2s).to(self.device)

    def npatch(self, gi):
        return (self.groups[gi][1] // self.patch) ** 2


_GEOM_CACHE = {}


def geometry(groups, device, patch=8):
    key = (tuple(groups), str(device), patch)
    if key not in _GEOM_CACHE:
        _GEOM_CACHE[key] = PackedGeometry(tuple(groups), patch, device)
    return _GEOM_CACHE[key]


# ----------------------------------------------------------------------------------------------- trunk descriptor
@dataclass
class TrunkSpec:
    """Where one pre-LN transformer trunk lives inside an arena (names are arena keys)."""
    dim: int
    heads: int
    mlp: int
    depth: int
    ln_eps: float
    attn_scale: float
    block_names: list                   # per block: dict(ln1_g=..., w_qkv=..., b_qkv=None|name, ...)

    @property
    def inner(self):
        return self.heads * 64


def make_trunk_desc(arena, spec: TrunkSpec, geom: PackedGeometry, drop_scales=None, with_grad=True, dropout_p=0.0, dropout_seed=0,
                    wgrad_overwrite=False, wgrad_workgroups=0, dropout_step=None, wgrad_defer=False):
    """Build the C descriptor (keeps the ctypes block array alive on the returned object)."""
    blocks = (_lib.BlockOffsets * spec.depth)()
    for i, nm in enumerate(spec.block_names):
        b = blocks[i]
        for f in ("ln1_g", "ln1_b", "w_qkv", "w_proj", "b_proj", "ln2_g", "ln2_b", "w_fc1", "b_fc1", "w_fc2", "b_fc2"):
            setattr(b, f, arena.offsets[nm[f]])
        b.b_qkv = arena.offsets[nm["b_qkv"]] if nm.get("b_qkv") else -1
        for f in ("w_qkv", "w_proj", "w_fc1", "w_fc2"):
            setattr(b, f + "_t", arena.t_offsets.get(nm[f], 0))      # teacher arenas keep no W^T (forward only)
    d = _lib.TrunkDesc()
    d.dim, d.inner, d.heads, d.mlp, d.depth = spec.dim, spec.inner, spec.heads, spec.mlp, spec.depth
    d.ln_eps, d.attn_scale = spec.ln_eps, spec.attn_scale
    d.n_tok, d.n_seq, d.max_len = geom.n_tok, geom.n_seq, geom.max_len
    d.cu_seqlens, d.row2seq = geom.cu_seqlens.data_ptr(), geom.row2seq.data_ptr()
    d.drop_scales = drop_scales.data_ptr() if drop_scales is not None else None
    d.dropout_p, d.dropout_seed = float(dropout_p), int(dropout_seed) & 0xFFFFFFFF
    d.dropout_step = dropout_step.data_ptr() if dropout_step is not None else None
    d.master, d.shadow, d.shadow_t = arena.master.data_ptr(), arena.shadow.data_ptr(), arena.shadow_t.data_ptr()
    d.grad = arena.grad.data_ptr() if (with_grad and arena.grad is not None) else None
    d.blocks = C.cast(blocks, C.POINTER(_lib.BlockOffsets))
    d.wgrad_overwrite = 1 if wgrad_overwrite else 0
    d.wgrad_defer = 1 if wgrad_defer else 0
    d.wgrad_workgroups = int(wgrad_workgroups)
    if 1 < len(geom.groups) <= 4:                        # one attention launch per crop resolution
        d.n_groups = len(geom.groups)
        for gi, (n_img, side) in enumerate(geom.groups):
            d.group_n_seq[gi], d.group_max_len[gi] = n_img, (side // geom.patch) ** 2 + 1
    ctx = getattr(arena, "ctx", None) or _lib.default_ctx(arena.master.device)
    d.ctx = ctx.handle
    d._keep = (blocks, drop_scales, geom, arena, dropout_step, ctx)
    return d


def trunk_workspace(desc, save, device):
    n = _lib.lib().lafs_trunk_workspace_bytes(C.byref(desc), 1 if save else 0)
    if n < 0:
        raise _lib.LafsHipError("lafs_trunk_workspace_bytes: " + _lib.lib().lafs_last_error().decode())
    return torch.empty(n, dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------------------------- ViT (DINO/timm style)
@dataclass
class ViTSpec:
    trunk: TrunkSpec
    prefix: str                         # arena key prefix of the VisionTransformer ('' or 'backbone.')
    patch_order: int = _lib.PATCH_ORDER_CHW
    w_patch: str = "patch_embed.proj.weight"
    b_patch: str = "patch_embed.proj.bias"
    cls: str = "cls_token"
    final_g: str = "norm.weight"
    final_b: str = "norm.bias"
    pos: str = "pos_embed"             # position table: resampled per crop size (DINO ViT) or sliced [:n+1] (Part-fViT)


class ViTState:
    """Buffers kept between forward and backward of one packed ViT pass."""
    __slots__ = ("geom", "desc", "ws", "x_in", "patches", "cls_rows", "stats", "feat", "dropout", "bwd_done")


EMB_DROP_SITE = 0x40000000          # seed offset of the embedding dropout (the trunk sites use seed + 3*layer + {0,1,2})


def vit_forward(arena, spec: ViTSpec, geom: PackedGeometry, imgs, pos_tokens, drop_scales=None, save=True,
                ws=None, x_in=None, x_out=None, dropout=None, wgrad_overwrite=False, wgrad_workgroups=0, wgrad_defer=False):
    """imgs: list of fp32 NCHW tensors (one per group); pos_tokens: list of fp32 [npatch+1, D] per group.
    dropout: None or (p_trunk, p_embedding, seed[, step]): element dropout of Part-fViT (counter-based masks, see lafs_hip.h);
    `step`: a DEVICE float tensor whose value x 7919 is added to the seed inside the kernels (graph-captured steps).
    Returns (feat f32 [n_seq, D], state)."""
    D = spec.trunk.dim
    dev = imgs[0].device
    pre = spec.prefix
    st = ViTState()
    st.geom = geom
    st.bwd_done = set()
    p_trunk, p_emb, dseed = dropout[:3] if dropout is not None else (0.0, 0.0, 0)
    dstep = dropout[3] if (dropout is not None and len(dropout) > 3) else None
    st.dropout = (p_trunk, p_emb, dseed, dstep)
    st.desc = make_trunk_desc(arena, spec.trunk, geom, drop_scales, with_grad=save, dropout_p=p_trunk, dropout_seed=dseed,
                              wgrad_overwrite=wgrad_overwrite, wgrad_workgroups=wgrad_workgroups, dropout_step=dstep, wgrad_defer=wgrad_defer)
    st.ws = ws if ws is not None else trunk_workspace(st.desc, save, dev)
    st.x_in = x_in if x_in is not None else torch.empty(geom.n_tok, D, device=dev, dtype=f32)
    if x_out is None:
        x_out = torch.empty(geom.n_tok, D, device=dev, dtype=f32)
    st.patches = []
    wpe = arena.bf(pre + spec.w_patch).view(D, -1)
    for gi, img in enumerate(imgs):
        n_img, np_ = geom.groups[gi][0], geom.npatch(gi)
        if img.dim() == 3:                                # already '(p1 p2 c)' patch vectors [n_img, n, 192] (3-D input path)
            pt = ops.scale_cast_bf16(img.reshape(-1, img.shape[-1]).contiguous().float())
        else:
            pt = ops.patchify(img, spec.patch_order)
        rows = st.x_in[geom.tok_start[gi]: geom.tok_start[gi] + n_img * (np_ + 1)]
        ops.gemm_nt(pt, wpe, _lib.EPI_EMBED_F32, bias=arena.view(arena.master, pre + spec.b_patch),
                    pos=pos_tokens[gi], npatch=np_, out=rows)
        call("lafs_embed_cls", _p(arena.view(arena.master, pre + spec.cls)), _p(pos_tokens[gi]), _p(rows), D, n_img, np_, D)
        st.patches.append(pt if save else None)
    if p_emb > 0:
        call("lafs_dropout_f32", _p(st.x_in), D, geom.n_tok, D, float(p_emb), (dseed + EMB_DROP_SITE) & 0xFFFFFFFF, _p(dstep))
    call("lafs_trunk_forward", C.byref(st.desc), _p(st.x_in), _p(x_out), _p(st.ws), 1 if save else 0)
    st.cls_rows = torch.empty(geom.n_seq, D, device=dev, dtype=f32)
    call("lafs_gather_cls", _p(x_out), D, _p(geom.cu_seqlens), geom.n_seq, D, _p(st.cls_rows))
    _, feat, st.stats = ops.layernorm_fwd(st.cls_rows, arena.view(arena.master, pre + spec.final_g),
                                          arena.view(arena.master, pre + spec.final_b), spec.trunk.ln_eps,
                                          want_bf16=False, want_f32=True)
    st.feat = feat
    return feat, st, x_out


def vit_backward_begin(arena, spec: ViTSpec, st: ViTState, dfeat, g_buf=None):
    """Final-LayerNorm backward on the cls rows and scatter into the (zeroed) residual-gradient buffer.  Returns g."""
    geom, D, pre = st.geom, spec.trunk.dim, spec.prefix
    dev = dfeat.device
    gv = lambda n: arena.view(arena.grad, pre + n)
    dcls_rows = torch.empty(geom.n_seq, D, device=dev, dtype=f32)
    ops.layernorm_bwd(dfeat.contiguous(), st.cls_rows, st.stats, arena.view(arena.master, pre + spec.final_g), dcls_rows,
                      gv(spec.final_g), gv(spec.final_b), accumulate=False)
    g = g_buf if g_buf is not None else torch.empty(geom.n_tok, D, device=dev, dtype=f32)
    ops.zero_(g)
    call("lafs_scatter_cls", _p(dcls_rows), _p(geom.cu_seqlens), geom.n_seq, D, _p(g), D)
    return g


def vit_backward_layers(st: ViTState, g, hi, lo, wgrad_stream=None):
    """Blocks hi-1 .. lo of the trunk backward (weight-gradient GEMMs optionally on `wgrad_stream`).
    With `wgrad_overwrite` the block weight gradients are WRITTEN, not accumulated (their zeroing is skipped, SEG_OVERWRITTEN): each
    block may be walked once per forward -- a second pass would silently replace the first one's gradients, so it is refused."""
    if st.desc.wgrad_overwrite:
        again = st.bwd_done.intersection(range(lo, hi))
        if again:
            raise _lib.LafsHipError(f"trunk backward over blocks {sorted(again)} twice with written (not accumulated) weight gradients")
        st.bwd_done.update(range(lo, hi))
    ws2 = C.c_void_p(wgrad_stream.cuda_stream) if wgrad_stream is not None else None
    call("lafs_trunk_backward", C.byref(st.desc), _p(st.x_in), _p(g), _p(st.ws), hi, lo, ws2)


def vit_wgrad_layers(st: ViTState, hi, lo):
    """The weight gradients of blocks hi-1 .. lo that a backward with `wgrad_defer` left out (lafs_trunk_wgrad), on the current
    stream: the caller places them beside whatever leaves the chip idle."""
    if not st.desc.wgrad_defer:
        raise _lib.LafsHipError("vit_wgrad_layers: the forward was not run with wgrad_defer")
    missing = set(range(lo, hi)) - st.bwd_done if st.desc.wgrad_overwrite else set()
    if missing:
        raise _lib.LafsHipError(f"deferred weight gradients of blocks {sorted(missing)} requested before their backward ran")
    call("lafs_trunk_wgrad", C.byref(st.desc), _p(st.ws), hi, lo)


def vit_backward_end(arena, spec: ViTSpec, st: ViTState, g, want_dx=False, dpos_out=None):
    """Token-assembly / patch-embedding backward.  Returns the list of dpos f32 [npatch+1, D] per group (the bicubic
    resampling of the position table lives in torch); with ``want_dx`` also the gradient of the patch vectors
    f32 [n_img, npatch, 192] per group (the landmark branch of Part-fViT differentiates through the patches,
    reference face_pre_pro/ViT_face.py:679-711)."""
    geom, D, pre = st.geom, spec.trunk.dim, spec.prefix
    dev = g.device
    gv = lambda n: arena.view(arena.grad, pre + n)
    dpos, dx = [], []
    p_emb, dseed, dstep = st.dropout[1], st.dropout[2], st.dropout[3]
    if p_emb > 0:                                         # backward of the embedding dropout: the same mask on the gradient
        call("lafs_dropout_f32", _p(g), D, geom.n_tok, D, float(p_emb), (dseed + EMB_DROP_SITE) & 0xFFFFFFFF, _p(dstep))
    wt = None
    if want_dx:                                           # W_patch^T bf16 [192, D]: B operand of dP = dTok @ W_patch
        wt = torch.empty(192, D, device=dev, dtype=bf16)
        call("lafs_transpose_cast_bf16", _p(arena.view(arena.master, pre + spec.w_patch)), D, 192, _p(wt), D)
    for gi in range(len(geom.groups)):
        n_img, np_ = geom.groups[gi][0], geom.npatch(gi)
        rows = g[geom.tok_start[gi]: geom.tok_start[gi] + n_img * (np_ + 1)]
        gp = torch.empty(n_img * np_, D, device=dev, dtype=bf16)
        # dpos_out[gi]: accumulate this group's position gradient straight into a caller buffer (e.g. the arena's gradient rows of
        # a table that needs no resampling) instead of a fresh zeroed temporary
        dp = dpos_out[gi] if (dpos_out is not None and dpos_out[gi] is not None) else ops.zeros(np_ + 1, D, device=dev)
        # position / cls sums and the patch embedding's weight + bias gradients WITHOUT fp32 atomics (round 6: per-chunk slots folded in
        # a fixed order; the wide-tile weight-gradient kernel with its slice fold) -- the last two atomic producers of the LAFS step:
        # the step is bit-reproducible run to run (tools/lab/chains_probe.sh)
        ws = arena.scratch(("embed_bwd", n_img, np_, D), int(_lib.lib().lafs_embed_bwd_workspace_bytes(n_img, np_, D)) // 4)
        call("lafs_embed_bwd", _p(rows), D, n_img, np_, D, _p(gp), _p(dp), _p(gv(spec.cls)), _p(ws))
        # (capped to EMBED_WGRAD_CUS workgroups: the output is two 192 x 192 tiles, and the default plan -- as many token slices as fill
        # the chip -- would fold 128 slice images, 38 MB, for a 295 KB gradient)
        prob = [(gp, st.patches[gi], gv(spec.w_patch).view(D, -1), True, gv(spec.b_patch))]
        items, Mp = ops._wgrad_items(prob)
        wws = arena.scratch(("embed_wgrad", Mp, D), max(int(_lib.lib().lafs_wgrad_group_workspace_bytes(items, 1, Mp, EMBED_WGRAD_CUS)), 16) // 4)
        ops.wgrad_group(prob, workspace=wws, max_workgroups=EMBED_WGRAD_CUS)
        dpos.append(dp)
        if want_dx:
            dx.append(ops.gemm_nt(gp, wt, _lib.EPI_F32).view(n_img, np_, 192))
    return (dpos, dx) if want_dx else dpos


EMBED_WGRAD_CUS = int(os.environ.get("LAFS_EMBED_WGRAD_CUS", "64"))      # lab knob (tools/lab/NOTES.md)


def unpatchify_grad(dpatch, order):
    """Patch-vector gradient [B, n, 192] -> image gradient [B, 3, 8r, 8r] (pure re-indexing; the inverse of lafs_patchify)."""
    B, n, _ = dpatch.shape
    r = int(math.isqrt(n))
    if order == _lib.PATCH_ORDER_HWC:                     # '(p1 p2 c)'
        return dpatch.view(B, r, r, 8, 8, 3).permute(0, 5, 1, 3, 2, 4).reshape(B, 3, 8 * r, 8 * r)
    return dpatch.view(B, r, r, 3, 8, 8).permute(0, 3, 1, 4, 2, 5).reshape(B, 3, 8 * r, 8 * r)


def vit_backward(arena, spec: ViTSpec, st: ViTState, dfeat, g_buf=None, wgrad_stream=None, want_dx=False):
    """dfeat f32 [n_seq, D].  Accumulates every parameter gradient into arena.grad EXCEPT the position table, whose
    per-group gradients are returned (plus the patch-vector gradients when ``want_dx``)."""
    g = vit_backward_begin(arena, spec, st, dfeat, g_buf)
    vit_backward_layers(st, g, spec.trunk.depth, 0, wgrad_stream)
    return vit_backward_end(arena, spec, st, g, want_dx)


# ----------------------------------------------------------------------------------------------- DINO head
class HeadState:
    __slots__ = ("x_bf", "u1", "a1", "u2", "a2", "z", "zn", "inv_z", "wn", "wn_t", "inv_v", "logits", "K", "Kpad")


def head_forward(arena, prefix, x, K, save=True, logits=None, skip_logits=False):
    """DINOHead: x f32 [n, in_dim] -> logits f32 [n, Kpad] (columns >= K are zero).  skip_logits: stop in front of the last GEMM
    (st.zn / st.wn are its operands): the training engine forms the logits inside the fused loss (ops.dino_head_loss)."""
    n = x.shape[0]
    dev = x.device
    st = HeadState()
    st.K, st.Kpad = K, (K + 127) // 128 * 128
    m = lambda k: arena.view(arena.master, prefix + k)
    st.x_bf = ops.scale_cast_bf16(x.contiguous())
    st.u1, st.a1 = ops.gemm_nt(st.x_bf, arena.bf(prefix + "mlp.0.weight"), _lib.EPI_BF16_GELU, bias=m("mlp.0.bias"))
    st.u2, st.a2 = ops.gemm_nt(st.a1, arena.bf(prefix + "mlp.2.weight"), _lib.EPI_BF16_GELU, bias=m("mlp.2.bias"))
    st.z = ops.gemm_nt(st.a2, arena.bf(prefix + "mlp.4.weight"), _lib.EPI_F32, bias=m("mlp.4.bias"))
    Db = st.z.shape[1]
    st.zn = torch.empty(n, Db, device=dev, dtype=bf16)
    st.inv_z = torch.empty(n, device=dev, dtype=f32)
    call("lafs_l2norm_fwd", _p(st.z), Db, _p(st.zn), Db, _p(st.inv_z), n, Db)
    st.wn = torch.empty(st.Kpad, Db, device=dev, dtype=bf16)
    st.wn_t = torch.empty(Db, st.Kpad, device=dev, dtype=bf16) if save else None
    st.inv_v = torch.empty(K, device=dev, dtype=f32)
    call("lafs_weightnorm_fwd", _p(m("last_layer.weight_v")), _p(m("last_layer.weight_g")), K, st.Kpad, Db, _p(st.wn),
         _p(st.wn_t), st.Kpad, _p(st.inv_v))
    if skip_logits:
        st.logits = None
        return None, st
    if logits is None:
        logits = torch.empty(n, st.Kpad, device=dev, dtype=f32)
    ops.gemm_nt(st.zn, st.wn, _lib.EPI_F32, out=logits, n_cols=st.Kpad)
    st.logits = logits
    return logits, st


def head_backward(arena, prefix, st: HeadState, dlogits_bf, train_g=False, overwrite_last=False):
    """dlogits_bf: bf16 [n, Kpad] (pad columns zero).  Accumulates into arena.grad; returns dx f32 [n, in_dim].
    overwrite_last: WRITE the last layer's weight_v (and weight_g) gradient instead of accumulating -- the training engine runs
    one backward per step and skips zeroing that 100 MB tensor (LAFS_SEG_OVERWRITTEN)."""
    n, dev = dlogits_bf.shape[0], dlogits_bf.device
    K, Kpad = st.K, st.Kpad
    Db = st.z.shape[1]
    m = lambda k: arena.view(arena.master, prefix + k)
    gv = lambda k: arena.view(arena.grad, prefix + k)
    p2 = lambda k: arena.params[arena.names.index(prefix + k)].shape
    # dzn = dlogits Wn: the class axis (K = 100 096) is the reduction -> K-split into slice images + one fold (no atomics:
    # 64 slices x 640 x 256 fp32 = 42 MB of plain stores instead of 10 M same-region atomics; 187 -> ~60 us at C2)
    splits = _lib.lib().lafs_gemm_nt_slices(Kpad, max(1, min(32, Kpad // 1024)))      # what the request really yields
    part = torch.empty(splits, dlogits_bf.shape[0], st.wn_t.shape[0], device=dev, dtype=f32)
    ops.gemm_nt(dlogits_bf, st.wn_t, _lib.EPI_F32, splits=splits, out=part.view(-1, part.shape[2]), out_rows=part.shape[0] * part.shape[1])
    dzn = torch.empty(part.shape[1], part.shape[2], device=dev, dtype=f32)
    call("lafs_sum_slices", _p(part), part.shape[1] * part.shape[2], splits, dzn.numel(), _p(dzn))
    # d(normalised weights) = dlogits^T zn: M = n rows is a single token slice of the wide-tile kernel -> written directly, no
    # zero-fill of the 100 MB buffer and no atomics
    dwn = torch.empty(Kpad, Db, device=dev, dtype=f32)
    ops.wgrad(dlogits_bf, st.zn, dwn, accumulate=False)
    dg = gv("last_layer.weight_g") if train_g else None
    call("lafs_weightnorm_bwd", _p(dwn), _p(m("last_layer.weight_v")), _p(m("last_layer.weight_g")), _p(st.inv_v), K, Db,
         _p(gv("last_layer.weight_v")), _p(dg), 0 if overwrite_last else 1)
    dz = torch.empty(n, Db, device=dev, dtype=f32)
    call("lafs_l2norm_bwd", _p(st.z), Db, _p(dzn), Db, _p(st.inv_z), _p(dz), Db, n, Db)
    dz_bf = ops.scale_cast_bf16(dz)
    du2 = ops.gemm_nt(dz_bf, arena.tview(prefix + "mlp.4.weight"), _lib.EPI_DGELU_BF16, aux=st.u2)
    du1 = ops.gemm_nt(du2, arena.tview(prefix + "mlp.2.weight"), _lib.EPI_DGELU_BF16, aux=st.u1)
    dx = ops.gemm_nt(du1, arena.tview(prefix + "mlp.0.weight"), _lib.EPI_F32)
    # the three MLP weight (+ bias) gradients as ONE grouped launch of the wide-tile kernel once all their operands exist
    # (three 128x128-tile launches of ~39 us each before: 640 token rows leave each of them a fraction of the chip)
    ops.wgrad_group([(dz_bf, st.a2, gv("mlp.4.weight").view(p2("mlp.4.weight")), True, gv("mlp.4.bias")),
                     (du2, st.a1, gv("mlp.2.weight").view(p2("mlp.2.weight")), True, gv("mlp.2.bias")),
                     (du1, st.x_bf, gv("mlp.0.weight").view(p2("mlp.0.weight")), True, gv("mlp.0.bias"))])
    return dx
