"""Tensor-level wrappers over the C ABI (torch is only the owner of device memory and streams here).

Every function launches hand-written HIP kernels from liblafs_hip.so on torch's current stream; nothing in this
module computes on the host or falls back to ATen math.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import call

bf16 = torch.bfloat16


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.LafsHipError(f"{name}: expected a device tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.LafsHipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if t.dim() >= 2 and t.stride(-1) != 1:
        raise _lib.LafsHipError(f"{name}: last dimension must be contiguous")


def _ld(t):
    return t.stride(0) if t.dim() >= 2 else t.numel()


def zero_(t):
    """t.zero_() as a lafs_fill_zero launch (keeps the captured step free of at::native kernels)."""
    if t.numel():
        call("lafs_fill_zero", _p(t), t.numel() * t.element_size())
    return t


def zeros(*shape, device, dtype=torch.float32):
    return zero_(torch.empty(*shape, device=device, dtype=dtype))


def gemm_nt(A, B, epilogue=_lib.EPI_BF16, bias=None, out=None, out2=None, resid=None, seq_scale=None, row2seq=None,
            aux=None, pos=None, npatch=0, splits=1, n_cols=None, out_rows=None, drop_p=0.0, drop_seed=0, act=0, skip_pre=False, drop_step=None, drop_row0=0,
            route_only=False, ctx=None):
    """out[M,N] = A[M,K] @ B[N,K]^T with a fused epilogue (see lafs_hip.h LAFS_EPI_*).  skip_pre (BF16_GELU): write only
    GELU(u), as the forward-only teacher pass does; route_only: return lafs_gemm_nt_route for this request instead of running it."""
    # 16-bit operand format: bf16 everywhere except the trainable landmark CNN's plan, which runs on fp16 (operand_f16)
    h16 = torch.float16 if A.dtype == torch.float16 else bf16
    _chk(A, h16, "A"); _chk(B, h16, "B")
    M, K = A.shape
    N = B.shape[0] if n_cols is None else n_cols
    f32_out = epilogue in (_lib.EPI_RESID_F32, _lib.EPI_F32, _lib.EPI_ATOMIC_F32, _lib.EPI_EMBED_F32)
    if out is None:
        rows = M if out_rows is None else out_rows
        out = torch.empty(rows, N, device=A.device, dtype=torch.float32 if f32_out else h16)
        if epilogue == _lib.EPI_ATOMIC_F32:
            zero_(out)
    _chk(out, torch.float32 if f32_out else h16, "out")
    if epilogue == _lib.EPI_BF16_GELU and out2 is None:
        out2 = torch.empty(M, N, device=A.device, dtype=bf16)
    a = _lib.GemmNTArgs()
    a.A, a.lda, a.B, a.ldb = A.data_ptr(), _ld(A), B.data_ptr(), _ld(B)
    a.M, a.N, a.K, a.epilogue = M, N, K, epilogue
    a.drop_p, a.drop_seed = float(drop_p), int(drop_seed) & 0xFFFFFFFF
    a.drop_step, a.drop_row0 = (drop_step.data_ptr() if drop_step is not None else None), int(drop_row0)
    a.act = int(act)
    a.operand_f16 = 1 if h16 == torch.float16 else 0
    a.C, a.ldc = out.data_ptr(), _ld(out)
    if out2 is not None:
        _chk(out2, bf16, "out2"); a.C2, a.ldc2 = out2.data_ptr(), _ld(out2)
    if bias is not None:
        _chk(bias, torch.float32, "bias"); a.bias = bias.data_ptr()
    if resid is not None:
        _chk(resid, torch.float32, "resid"); a.resid, a.ldr = resid.data_ptr(), _ld(resid)
    if seq_scale is not None:
        _chk(seq_scale, torch.float32, "seq_scale"); _chk(row2seq, torch.int32, "row2seq")
        a.seq_scale, a.row2seq = seq_scale.data_ptr(), row2seq.data_ptr()
    if aux is not None:
        _chk(aux, h16, "aux"); a.aux, a.ldaux = aux.data_ptr(), _ld(aux)
    if pos is not None:
        _chk(pos, torch.float32, "pos"); a.pos, a.npatch = pos.data_ptr(), npatch
    a.splits = splits
    a.ctx = (ctx or _lib.default_ctx(A.device)).handle
    if skip_pre and epilogue == _lib.EPI_BF16_GELU:
        a.C = None
    if route_only:
        return int(_lib.lib().lafs_gemm_nt_route(C.byref(a)))
    call("lafs_gemm_nt", C.byref(a))
    return (out, out2) if epilogue == _lib.EPI_BF16_GELU else out


def mlp_fused(X, Wa, Wb, mode, bias_a=None, bias_b=None, resid=None, seq_scale=None, row2seq=None, out=None, save_grad=None, save_act=None, ctx=None,
              ln=None, ln_stats=None, ln_out=None, ln_bwd=None, next_ln=None, proj=None):
    """The fused MLP of a ViT-S block (lafs_mlp_fused, csrc/mlp_fused.hip; vision_transformer.py:59-65,112).
    MLP_FWD / MLP_FWD_SAVE: out(f32) = resid + seq_scale[row2seq] * (gelu(X Wa^T + bias_a) Wb^T + bias_b), the saving form also
    writes save_grad = gelu'(u) and save_act = gelu(u); MLP_BWD: save_act = du = (X Wa^T) * save_grad, out(bf16) = du Wb^T."""
    _chk(Wa, bf16, "Wa"); _chk(Wb, bf16, "Wb")
    if ln is None:
        _chk(X, bf16, "X")
    else:
        X = resid                                        # ln = (gamma, beta, eps): the operand is LayerNorm(resid), formed in the kernel
    M, H = X.shape[0], Wa.shape[0]
    fwd = mode != _lib.MLP_BWD
    if out is None:
        out = torch.empty(M, X.shape[1], device=X.device, dtype=torch.float32 if fwd else bf16)
    _chk(out, torch.float32 if fwd else bf16, "out")
    if mode == _lib.MLP_FWD_SAVE and save_grad is None:
        save_grad = torch.empty(M, H, device=X.device, dtype=bf16)
    if mode != _lib.MLP_FWD and save_act is None:
        save_act = torch.empty(M, H, device=X.device, dtype=bf16)
    a = _lib.MlpArgs()
    a.X, a.ldx, a.Wa, a.ldwa, a.Wb, a.ldwb = (X.data_ptr() if ln is None else None), _ld(X), Wa.data_ptr(), _ld(Wa), Wb.data_ptr(), _ld(Wb)
    a.M, a.H, a.mode = M, H, int(mode)
    if ln is not None:
        _chk(ln[0], torch.float32, "ln gamma"); _chk(ln[1], torch.float32, "ln beta")
        a.ln_gamma, a.ln_beta, a.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2])
        if ln_stats is not None:
            _chk(ln_stats, torch.float32, "ln_stats"); a.ln_stats = ln_stats.data_ptr()
        if ln_out is not None:
            _chk(ln_out, bf16, "ln_out"); a.ln_out, a.ldln = ln_out.data_ptr(), _ld(ln_out)
    if bias_a is not None:
        _chk(bias_a, torch.float32, "bias_a"); a.bias_a = bias_a.data_ptr()
    if bias_b is not None:
        _chk(bias_b, torch.float32, "bias_b"); a.bias_b = bias_b.data_ptr()
    if resid is not None:
        _chk(resid, torch.float32, "resid"); a.resid, a.ldr = resid.data_ptr(), _ld(resid)
    if seq_scale is not None:
        _chk(seq_scale, torch.float32, "seq_scale"); _chk(row2seq, torch.int32, "row2seq")
        a.seq_scale, a.row2seq = seq_scale.data_ptr(), row2seq.data_ptr()
    a.out, a.ldo = out.data_ptr(), _ld(out)
    if save_grad is not None:
        _chk(save_grad, bf16, "save_grad"); a.save_grad, a.ldsg = save_grad.data_ptr(), _ld(save_grad)
    if save_act is not None:
        _chk(save_act, bf16, "save_act"); a.save_act, a.ldsa = save_act.data_ptr(), _ld(save_act)
    if next_ln is not None:                              # forward modes: (gamma, beta, eps, out bf16, stats or None) -- the next block's LayerNorm 1
        ng, nb, ne, no_, ns = next_ln
        _chk(ng, torch.float32, "next gamma"); _chk(nb, torch.float32, "next beta"); _chk(no_, bf16, "next_ln out")
        a.next_ln_gamma, a.next_ln_beta, a.next_ln_eps, a.next_ln_out, a.ldnln_next = ng.data_ptr(), nb.data_ptr(), float(ne), no_.data_ptr(), _ld(no_)
        if ns is not None:
            _chk(ns, torch.float32, "next_ln stats"); a.next_ln_stats = ns.data_ptr()
    if proj is not None:                                 # forward modes with ln: (o bf16, Wp bf16 [384, 384], bias or None, resid0 f32, scale or None) -- resid is WRITTEN
        po, pw, pb, pr, ps = proj
        _chk(po, bf16, "proj o"); _chk(pw, bf16, "proj W"); _chk(pr, torch.float32, "proj resid")
        a.proj_x, a.ldpx, a.proj_w, a.ldpw, a.proj_resid, a.ldpr = po.data_ptr(), _ld(po), pw.data_ptr(), _ld(pw), pr.data_ptr(), _ld(pr)
        if pb is not None:
            _chk(pb, torch.float32, "proj bias"); a.proj_bias = pb.data_ptr()
        if ps is not None:
            _chk(ps, torch.float32, "proj scale"); a.proj_scale = ps.data_ptr()
    if ln_bwd is not None:                               # MLP_BWD: (x, stats, gamma, g_io, gb_out, part_out) -- LayerNorm backward in the epilogue
        x_, st_, gam_, gio_, gbo_, part_ = ln_bwd
        _chk(x_, torch.float32, "x"); _chk(st_, torch.float32, "stats"); _chk(gam_, torch.float32, "gamma"); _chk(gio_, torch.float32, "g_io")
        _chk(gbo_, bf16, "gb_out"); _chk(part_, torch.float32, "part_out")
        a.resid, a.ldr, a.ln_stats, a.ln_gamma = x_.data_ptr(), _ld(x_), st_.data_ptr(), gam_.data_ptr()
        a.ln_g_io, a.ldgio, a.ln_gb_out, a.ldgb, a.ln_part_out = gio_.data_ptr(), _ld(gio_), gbo_.data_ptr(), _ld(gbo_), part_.data_ptr()
    a.ctx = (ctx or _lib.default_ctx(X.device)).handle
    call("lafs_mlp_fused", C.byref(a))
    return out, save_grad, save_act


def gemm_tn_acc(A, B, Cacc, splits=0, colsum=None):
    """Cacc[N1,N2] (f32) += A[M,N1]^T @ B[M,N2]; optional colsum[N1] (f32) += column sums of A (bias gradient)."""
    _chk(A, bf16, "A"); _chk(B, bf16, "B"); _chk(Cacc, torch.float32, "C")
    M, N1 = A.shape
    N2 = B.shape[1]
    call("lafs_gemm_tn_acc", _p(A), _ld(A), _p(B), _ld(B), _p(Cacc), _ld(Cacc), M, N1, N2, splits, _p(colsum))
    return Cacc


def wgrad_workspace(M, N1, N2, device):
    n = int(_lib.lib().lafs_wgrad_workspace_bytes(M, N1, N2))
    if n < 0:
        raise _lib.LafsHipError("lafs_wgrad_workspace_bytes: bad shape")
    return torch.empty(max(n, 16) // 4, device=device, dtype=torch.float32)


def wgrad(A, B, C, accumulate=True, colsum=None, workspace=None):
    """C[N1,N2] (f32) = (accumulate ? C : 0) + A[M,N1]^T @ B[M,N2] on the wide-tile kernel (csrc/wgrad.hip)."""
    h16 = torch.float16 if A.dtype == torch.float16 else bf16
    _chk(A, h16, "A"); _chk(B, h16, "B"); _chk(C, torch.float32, "C")
    M, N1 = A.shape
    N2 = B.shape[1]
    if workspace is None:
        workspace = wgrad_workspace(M, N1, N2, A.device)
    call("lafs_wgrad_f16" if h16 == torch.float16 else "lafs_wgrad", _p(A), _ld(A), _p(B), _ld(B), _p(C), _ld(C), M, N1, N2, int(bool(accumulate)), _p(colsum),
         _p(workspace), workspace.numel() * workspace.element_size())
    return C


def _wgrad_items(problems):
    items = (_lib.WgradItem * len(problems))()
    M = problems[0][0].shape[0]
    for it, (A, B, Cacc, accumulate, colsum) in zip(items, problems):
        _chk(A, bf16, "A"); _chk(B, bf16, "B"); _chk(Cacc, torch.float32, "C")
        if A.shape[0] != M or B.shape[0] != M:
            raise _lib.LafsHipError("wgrad_group: all GEMMs of a group share the token count M")
        it.A, it.lda, it.B, it.ldb, it.C, it.ldc = A.data_ptr(), _ld(A), B.data_ptr(), _ld(B), Cacc.data_ptr(), _ld(Cacc)
        it.N1, it.N2, it.accumulate = A.shape[1], B.shape[1], int(bool(accumulate))
        it.colsum_a = None if colsum is None else colsum.data_ptr()
    return items, M


def wgrad_group(problems, workspace=None, max_workgroups=0):
    """problems: list of (A[M,N1] bf16, B[M,N2] bf16, C[N1,N2] f32, accumulate, colsum or None); one launch + one fold."""
    items, M = _wgrad_items(problems)
    if workspace is None:
        n = int(_lib.lib().lafs_wgrad_group_workspace_bytes(items, len(problems), M, max_workgroups))
        workspace = torch.empty(max(n, 16) // 4, device=problems[0][0].device, dtype=torch.float32)
    call("lafs_wgrad_group", items, len(problems), M, max_workgroups, _p(workspace), workspace.numel() * workspace.element_size())
    return workspace


def gemm_tn_part(A, B, part, splits=0, colsum=None):
    """part[x] += (A^T @ B restricted to the rows handled by XCD x); part f32 [N_XCD, N1, N2] (zero-initialised)."""
    _chk(A, bf16, "A"); _chk(B, bf16, "B"); _chk(part, torch.float32, "part")
    M, N1 = A.shape
    N2 = B.shape[1]
    call("lafs_gemm_tn_part", _p(A), _ld(A), _p(B), _ld(B), _p(part), part.stride(1), part.stride(0), M, N1, N2, splits, _p(colsum))
    return part


def reduce_partials(part, out):
    """out += part.sum(0); part = 0."""
    call("lafs_reduce_partials", _p(part), part.stride(0), part.shape[0], out.numel(), _p(out))
    return out


def colsum_bf16_acc(X, out):
    _chk(X, bf16, "X"); _chk(out, torch.float32, "out")
    call("lafs_colsum_bf16_acc", _p(X), _ld(X), X.shape[0], X.shape[1], _p(out))
    return out


def layernorm_fwd(x, gamma, beta, eps, want_bf16=True, want_f32=False):
    _chk(x, torch.float32, "x")
    rows, D = x.shape
    y = torch.empty(rows, D, device=x.device, dtype=bf16) if want_bf16 else None
    yf = torch.empty(rows, D, device=x.device, dtype=torch.float32) if want_f32 else None
    stats = torch.empty(rows, 2, device=x.device, dtype=torch.float32)
    call("lafs_layernorm_fwd", _p(x), _ld(x), _p(gamma), _p(beta), eps, _p(y), D, _p(yf), D, _p(stats), rows, D)
    return y, yf, stats


def layernorm_bwd(dy, x, stats, gamma, g_io, dgamma, dbeta, accumulate=True, gb_out=None, seq_scale=None, row2seq=None,
                  drop_p=0.0, drop_seed=0, drop_step=None, drop_row0=0):
    """dy: bf16 or f32 [rows, D].  g_io (f32) receives (accumulates) dx; returns g_io.  dgamma / dbeta += the parameter gradients,
    deterministically: the launch stores per-workgroup sums (part_out) and lafs_layernorm_bwd_fold adds them in a fixed order."""
    rows, D = x.shape
    dy_b = dy if dy.dtype == bf16 else None
    dy_f = dy if dy.dtype == torch.float32 else None
    n_parts = int(_lib.lib().lafs_layernorm_bwd_parts(rows, D))
    part = torch.empty(n_parts * 2 * D, device=x.device, dtype=torch.float32)
    call("lafs_layernorm_bwd", _p(dy_b), D if dy_b is None else _ld(dy_b), _p(dy_f), D if dy_f is None else _ld(dy_f),
         _p(x), _ld(x), _p(stats), _p(gamma), _p(g_io), _ld(g_io), 1 if accumulate else 0,
         _p(gb_out), D if gb_out is None else _ld(gb_out), _p(seq_scale), _p(row2seq), _p(dgamma), _p(dbeta), rows, D,
         float(drop_p), int(drop_seed) & 0xFFFFFFFF, _p(drop_step), int(drop_row0), _p(part))
    item = (_lib.LnFoldItem * 1)()
    item[0].part[0], item[0].n_parts[0] = part.data_ptr(), n_parts
    item[0].dgamma, item[0].dbeta = dgamma.data_ptr(), dbeta.data_ptr()
    call("lafs_layernorm_bwd_fold", item, 1, D)
    return g_io


def scale_cast_bf16(g, seq_scale=None, row2seq=None, out=None, drop_p=0.0, drop_seed=0, drop_step=None, drop_row0=0):
    rows, D = g.shape
    if out is None:
        out = torch.empty(rows, D, device=g.device, dtype=bf16)
    call("lafs_scale_cast_bf16", _p(g), _ld(g), _p(out), _ld(out), _p(seq_scale), _p(row2seq), rows, D,
         float(drop_p), int(drop_seed) & 0xFFFFFFFF, _p(drop_step), int(drop_row0))
    return out


def dropout_mask(rows, cols, p, seed, device="cuda"):
    """The factor matrix (0 or 1/(1-p)) the kernels apply for (seed, rows x cols) -- test/debug helper."""
    out = torch.empty(rows, cols, device=device, dtype=torch.float32)
    call("lafs_debug_dropout_mask", rows, cols, float(p), int(seed) & 0xFFFFFFFF, _p(out))
    return out


def attention_fwd(qkv, cu_seqlens, max_len, heads, scale):
    _chk(qkv, bf16, "qkv"); _chk(cu_seqlens, torch.int32, "cu_seqlens")
    T = qkv.shape[0]
    inner = heads * 64
    out = torch.empty(T, inner, device=qkv.device, dtype=bf16)
    lse = torch.empty(T, heads, device=qkv.device, dtype=torch.float32)
    call("lafs_attention_fwd", _p(qkv), _ld(qkv), _p(cu_seqlens), cu_seqlens.numel() - 1, max_len, heads, scale,
         _p(out), inner, _p(lse))
    return out, lse


def attention_bwd(qkv, out, dout, lse, cu_seqlens, max_len, heads, scale):
    _chk(dout, bf16, "dout")
    dqkv = torch.empty_like(qkv)
    call("lafs_attention_bwd", _p(qkv), _ld(qkv), _p(out), _ld(out), _p(dout), _ld(dout), _p(lse),
         _p(cu_seqlens), cu_seqlens.numel() - 1, max_len, heads, scale, _p(dqkv), _ld(dqkv))
    return dqkv


def patchify(img, order=_lib.PATCH_ORDER_CHW):
    _chk(img, torch.float32, "img")
    B, _, S, _ = img.shape
    out = torch.empty(B * (S // 8) ** 2, 192, device=img.device, dtype=bf16)
    call("lafs_patchify", _p(img.contiguous()), B, S, order, _p(out))
    return out


def dino_head_loss(zn_s, zn_t, wn_s, wn_t, center, ncrops, K, student_temp, teacher_temp, grad, loss=None, colsum=None, ws=None,
                   dev_temps=None, grad_scale=1.0):
    """DINOHead's last layer of both networks + DINO loss + centre row sums with the logits never stored (csrc/dino_head_loss.hip).
    zn_*: bf16 [rows, 256], wn_*: bf16 [Kpad, 256]; grad: bf16 [ncrops B, >= Kpad] (written); returns (loss[1], grad)."""
    for t, n in ((zn_s, "zn_s"), (zn_t, "zn_t"), (wn_s, "wn_s"), (wn_t, "wn_t"), (grad, "grad")):
        _chk(t, bf16, n)
    rows, B = zn_s.shape[0], zn_t.shape[0] // 2
    if rows != ncrops * B or zn_s.shape[1] != zn_t.shape[1] or wn_s.shape != wn_t.shape or not zn_s.is_contiguous() or not wn_s.is_contiguous() \
            or not zn_t.is_contiguous() or not wn_t.is_contiguous():
        raise _lib.LafsHipError("dino_head_loss: student rows = ncrops * B, teacher rows = 2 * B, contiguous operands of equal widths")
    if ws is None:
        ws = torch.empty(_lib.lib().lafs_dino_head_loss_workspace(ncrops, B, K), device=zn_s.device, dtype=torch.float32)
    if loss is None:
        loss = torch.empty(1, device=zn_s.device, dtype=torch.float32)
    call("lafs_dino_head_loss", _p(zn_s), _p(zn_t), _p(wn_s), _p(wn_t), zn_s.shape[1], _p(center), ncrops, B, K, wn_s.shape[0],
         student_temp, teacher_temp, _p(dev_temps), _p(loss), _p(grad), _ld(grad), grad_scale, _p(colsum), _p(ws))
    return loss, grad


def dino_loss_fwd_bwd(student, teacher, center, ncrops, student_temp, teacher_temp, K=None, grad=None, grad_bf16=True,
                      grad_scale=1.0, ws=None, loss=None, dev_temps=None):
    """Returns (loss[1] f32, grad [rows, ld] bf16|f32).  student/teacher may be padded (ld >= K)."""
    _chk(student, torch.float32, "student"); _chk(teacher, torch.float32, "teacher")
    rows, ld = student.shape[0], _ld(student)
    K = student.shape[1] if K is None else K
    B = rows // ncrops
    if ws is None:
        ws = torch.empty(_lib.lib().lafs_dino_loss_workspace(ncrops, B, K), device=student.device, dtype=torch.float32)
    if loss is None:
        loss = torch.empty(1, device=student.device, dtype=torch.float32)
    if grad is None:
        grad = torch.zeros(rows, ld, device=student.device, dtype=bf16 if grad_bf16 else torch.float32)
    call("lafs_dino_loss_fwd_bwd", _p(student), _p(teacher), ld, _p(center), ncrops, B, K, student_temp, teacher_temp,
         _p(loss), _p(grad), _ld(grad), 1 if grad.dtype == bf16 else 0, grad_scale, _p(ws), _p(dev_temps))
    return loss, grad
