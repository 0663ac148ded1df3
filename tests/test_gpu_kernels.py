"""GPU parity of the individual HIP kernels (through the C ABI) against fp32 CPU math on the same inputs."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from lafs_cvpr2024_amd import _lib, ops  # noqa: E402
from lafs_cvpr2024_amd.ops import _p, call  # noqa: E402

DEV = "cuda"
bf16 = torch.bfloat16


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def rnd_bf(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(bf16)


def test_lds_transpose_read_model():
    """lane l, slot j must receive element (l>>4)*64 + j*16 + (l&15) of a lane-linear ramp."""
    src = torch.arange(512, dtype=torch.int16, device=DEV)
    out = torch.zeros(256, dtype=torch.int16, device=DEV)
    _lib.call("lafs_debug_tr16", C.c_void_p(src.data_ptr()), C.c_void_p(out.data_ptr()))
    got = out.cpu().view(64, 4)
    exp = torch.tensor([[(l >> 4) * 64 + j * 16 + (l & 15) for j in range(4)] for l in range(64)], dtype=torch.int16)
    assert torch.equal(got, exp), f"transpose-read model mismatch:\n{got[:20]}"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 128), (197 * 3, 1152, 384), (640, 1000, 256),
                                   (128 * 196 + 7, 384, 1536)])      # the last one: 128x384 / 12-wave tiles for the plain and residual epilogues
def test_gemm_nt_epilogues(M, N, K):
    A, B = rnd_bf(M, K, seed=1), rnd_bf(N, K, scale=0.1, seed=2)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
    ref = A.float() @ B.float().t() + bias
    Ad, Bd, bd = A.to(DEV), B.to(DEV), bias.to(DEV)
    out = ops.gemm_nt(Ad, Bd, _lib.EPI_BF16, bias=bd)
    assert relerr(out.float(), ref) < 1e-2
    out = ops.gemm_nt(Ad, Bd, _lib.EPI_F32, bias=bd)
    assert relerr(out, ref) < 2e-5 * math.sqrt(K)
    u, a = ops.gemm_nt(Ad, Bd, _lib.EPI_BF16_GELU, bias=bd)
    assert relerr(u.float(), ref) < 1e-2 and relerr(a.float(), F.gelu(ref)) < 1e-2
    # residual + per-sequence DropPath scale
    nseq = 7
    row2seq = (torch.arange(M) * nseq // M).int()
    sc = torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.9, 0.0, 1.0 / 0.9, 1.0 / 0.9, 1.0 / 0.9])
    resid = torch.randn(M, N, generator=torch.Generator().manual_seed(4))
    out = ops.gemm_nt(Ad, Bd, _lib.EPI_RESID_F32, bias=bd, resid=resid.to(DEV), seq_scale=sc.to(DEV), row2seq=row2seq.to(DEV))
    assert relerr(out, resid + sc[row2seq.long()].unsqueeze(1) * ref) < 1e-4
    # dGELU
    aux = rnd_bf(M, N, seed=5)
    x = aux.float().requires_grad_(True)
    F.gelu(x).sum().backward()
    out = ops.gemm_nt(Ad, Bd, _lib.EPI_DGELU_BF16, aux=aux.to(DEV))
    assert relerr(out.float(), (A.float() @ B.float().t()) * x.grad) < 1e-2
    # LAFS_GELU_SAVE_GRAD: the forward stores gelu'(u) in place of u, the backward multiplies it in as it is
    xr = ref.clone().requires_grad_(True)
    F.gelu(xr).sum().backward()
    ug, a2 = ops.gemm_nt(Ad, Bd, _lib.EPI_BF16_GELU, bias=bd, act=1)
    assert relerr(ug.float(), xr.grad) < 1e-2 and torch.equal(a2, a)
    out2 = ops.gemm_nt(Ad, Bd, _lib.EPI_DGELU_BF16, aux=ug, act=1)
    assert relerr(out2.float(), (A.float() @ B.float().t()) * ug.float().cpu()) < 1e-2
    # split-K atomics
    out = ops.gemm_nt(Ad, Bd, _lib.EPI_ATOMIC_F32, splits=2)
    assert relerr(out, A.float() @ B.float().t()) < 2e-5 * math.sqrt(K)


@pytest.mark.parametrize("M,N,K", [(128 * 103 + 37, 640, 640), (128 * 103 + 37, 704, 768), (25216, 768, 2048)])
def test_gemm_nt_tall_tiles(M, N, K):
    """Long reductions whose 128-row tiles would spill into one more round of the 512 workgroup slots run on 160-row tiles (route 4).
    Every epilogue of that form -- plain, GELU pair (u / gelu'(u) first tensor), residual + DropPath scale + element dropout, GELU' --
    against fp32 torch, and BIT-IDENTICAL to the 128-row kernel on a row prefix small enough to take that route (same k order per
    output element); rows past M stay untouched."""
    A, B = rnd_bf(M, K, seed=21), rnd_bf(N, K, scale=0.1, seed=22)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(23))
    ref = A.float() @ B.float().t() + bias
    Ad, Bd, bd = A.to(DEV), B.to(DEV), bias.to(DEV)
    Ms = 2048 + 5
    want = (4, 5) if (M, N, K) == (25216, 768, 2048) else (4,)         # (the plain epilogue of that shape: 160x256 persistent tiles since round 5)
    assert ops.gemm_nt(Ad, Bd, _lib.EPI_BF16, bias=bd, route_only=True) in want, "expected the 160-row tile route"
    assert ops.gemm_nt(Ad[:Ms], Bd, _lib.EPI_BF16, bias=bd, route_only=True) == 0
    guard = 7.0

    def run(epi, rows, **kw):
        f32 = epi == _lib.EPI_RESID_F32
        out = torch.full((rows + 192, N), guard, device=DEV, dtype=torch.float32 if f32 else torch.bfloat16)
        kw = {k: (v[:rows] if (torch.is_tensor(v) and v.dim() == 2 and v.shape[0] == M) else v) for k, v in kw.items()}
        if "row2seq" in kw:
            kw["row2seq"] = kw["row2seq"][:rows]
        res = ops.gemm_nt(Ad[:rows], Bd, epi, out=out[:rows], **kw)
        assert float((out[rows:].float() - guard).abs().max()) == 0.0, "rows past M were written"
        return res

    for epi, kw, check in (
            (_lib.EPI_BF16, dict(bias=bd), lambda o: relerr(o.float(), ref) < 1e-2),
            (_lib.EPI_DGELU_BF16, dict(aux=rnd_bf(M, N, seed=25).to(DEV), act=1), None)):
        full, sub = run(epi, M, **kw), run(epi, Ms, **kw)
        assert torch.equal(full[:Ms], sub)
        if check is not None:
            assert check(full)
    # GELU pair, both first-tensor forms
    for act in (0, 1):
        o2f = torch.empty(M, N, device=DEV, dtype=torch.bfloat16); o2s = torch.empty(Ms, N, device=DEV, dtype=torch.bfloat16)
        uf = run(_lib.EPI_BF16_GELU, M, bias=bd, out2=o2f, act=act)[0]
        us = run(_lib.EPI_BF16_GELU, Ms, bias=bd, out2=o2s, act=act)[0]
        assert torch.equal(uf[:Ms], us) and torch.equal(o2f[:Ms], o2s)
        assert relerr(o2f.float(), F.gelu(ref)) < 1e-2
        if act == 0:
            assert relerr(uf.float(), ref) < 1e-2
    # residual + DropPath scale + element dropout
    nseq = 7
    row2seq = (torch.arange(M) * nseq // M).int().to(DEV)
    sc = torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.9, 0.0, 1.0 / 0.9, 1.0 / 0.9, 1.0 / 0.9]).to(DEV)
    resid = torch.randn(M, N, generator=torch.Generator().manual_seed(24)).to(DEV)
    for p_drop in (0.0, 0.1):
        kw = dict(bias=bd, resid=resid, seq_scale=sc, row2seq=row2seq, drop_p=p_drop, drop_seed=77)
        full, sub = run(_lib.EPI_RESID_F32, M, **kw), run(_lib.EPI_RESID_F32, Ms, **kw)
        assert torch.equal(full[:Ms], sub)
        if p_drop == 0.0:
            assert relerr(full, resid.cpu() + sc.cpu()[row2seq.cpu().long()].unsqueeze(1) * ref) < 1e-4


@pytest.mark.parametrize("M,N", [(2048 + 77, 384), (4096 + 5, 1152), (2560, 1536), (128 * 41 + 1, 1536), (128 * 70 + 33, 384)])
def test_gemm_nt_k_resident_kernel(M, N):
    """The K = 384 streaming shapes of the ViT-S trunk run on the K-resident kernel (gemm_kres.hip): every epilogue it covers
    against fp32 torch, ragged row counts (last row unit partly / wholly beyond M for some waves), in-place residual."""
    K = 384
    route = 1
    A, B = rnd_bf(M, K, seed=11), rnd_bf(N, K, scale=0.1, seed=12)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(13))
    ref = A.float() @ B.float().t() + bias
    Ad, Bd, bd = A.to(DEV), B.to(DEV), bias.to(DEV)
    assert ops.gemm_nt(Ad, Bd, _lib.EPI_BF16, bias=bd, route_only=True) == route, "expected the K-resident route"
    assert ops.gemm_nt(Ad[:1024], Bd, _lib.EPI_BF16, bias=bd, route_only=True) == 0
    guard = 7.0
    out = torch.full((M + 64, N), guard, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(Ad, Bd, _lib.EPI_BF16, bias=bd, out=out[:M])
    assert relerr(out[:M].float(), ref) < 1e-2
    assert torch.all(out[M:] == guard), "rows beyond M were written"
    u, a = ops.gemm_nt(Ad, Bd, _lib.EPI_BF16_GELU, bias=bd)
    assert relerr(u.float(), ref) < 1e-2 and relerr(a.float(), F.gelu(ref)) < 1e-2
    u2 = torch.full((M, N), guard, device=DEV, dtype=torch.bfloat16)
    _, a2 = ops.gemm_nt(Ad, Bd, _lib.EPI_BF16_GELU, bias=bd, out=u2, skip_pre=True)
    assert torch.equal(a2, a) and torch.all(u2 == guard), "forward-only pass: same GELU(u), no pre-activation store"
    nseq = 7
    row2seq = (torch.arange(M) * nseq // M).int()
    sc = torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.9, 0.0, 1.0 / 0.9, 1.0 / 0.9, 1.0 / 0.9])
    resid = torch.randn(M, N, generator=torch.Generator().manual_seed(14))
    exp = resid + sc[row2seq.long()].unsqueeze(1) * ref
    assert ops.gemm_nt(Ad, Bd, _lib.EPI_RESID_F32, bias=bd, resid=resid.to(DEV), route_only=True) == route
    out = ops.gemm_nt(Ad, Bd, _lib.EPI_RESID_F32, bias=bd, resid=resid.to(DEV), seq_scale=sc.to(DEV), row2seq=row2seq.to(DEV))
    assert relerr(out, exp) < 2e-4
    inplace = resid.to(DEV)
    ops.gemm_nt(Ad, Bd, _lib.EPI_RESID_F32, bias=bd, resid=inplace, seq_scale=sc.to(DEV), row2seq=row2seq.to(DEV), out=inplace)
    assert torch.equal(inplace, out), "in-place residual"
    aux = rnd_bf(M, N, seed=15)
    x = aux.float().requires_grad_(True)
    F.gelu(x).sum().backward()
    assert ops.gemm_nt(Ad, Bd, _lib.EPI_DGELU_BF16, aux=aux.to(DEV), route_only=True) == route
    out = ops.gemm_nt(Ad, Bd, _lib.EPI_DGELU_BF16, aux=aux.to(DEV))
    assert relerr(out.float(), (A.float() @ B.float().t()) * x.grad) < 1e-2
    # LAFS_GELU_SAVE_GRAD (what the trunk uses): gelu'(u) saved by the forward, multiplied in by the backward
    xr = ref.clone().requires_grad_(True)
    F.gelu(xr).sum().backward()
    ug, a3 = ops.gemm_nt(Ad, Bd, _lib.EPI_BF16_GELU, bias=bd, act=1)
    assert relerr(ug.float(), xr.grad) < 1e-2 and relerr(a3.float(), F.gelu(ref)) < 1e-2
    out = ops.gemm_nt(Ad, Bd, _lib.EPI_DGELU_BF16, aux=ug, act=1)
    assert relerr(out.float(), (A.float() @ B.float().t()) * ug.float().cpu()) < 1e-2
    # a K split in two slices keeps a request off the K-resident route
    assert ops.gemm_nt(Ad, Bd, _lib.EPI_F32, splits=2, route_only=True) == 0


@pytest.mark.parametrize("geo", [2, 3, 4, 5])
@pytest.mark.parametrize("M,N,K", [(192 * 75 + 37, 768, 2048), (192 * 54 + 5, 2112, 768), (192 * 80 + 37, 704, 768), (12288, 2048, 512)])
def test_gemm_nt_big_tiles_equal_the_tiled_kernel_bit_for_bit(M, N, K, geo):
    """The wide long-K linears of the Part-fViT trunk (FeedForward 768 <-> 2048, to_qkv 2112, to_out 704; face_pre_pro/ViT_face.py:
    126-149) on the 192x256 one-workgroup-per-CU kernel (gemm_big.hip, route 5) against the 128x128 tiled kernel selected through a
    context with LAFS_OPT_NT_BIG = 0: the same MFMA, the same k order per output element, the same epilogue arithmetic -- plain,
    GELU pair (both first-tensor forms, forward-only form), residual + DropPath scale, GELU', each with and without element dropout --
    must agree BIT FOR BIT; against fp32 torch within the bf16 tolerances; ragged last row tile (rows past M untouched), ragged last
    column tile (N = 704, 2112), many tiles per persistent workgroup (the operand ring runs across tile boundaries)."""
    A, B = rnd_bf(M, K, seed=31), rnd_bf(N, K, scale=0.05, seed=32)
    g = torch.Generator().manual_seed(33)
    bias = torch.randn(N, generator=g)
    ref = A.float() @ B.float().t() + bias
    Ad, Bd, bd = A.to(DEV), B.to(DEV), bias.to(DEV)
    tiled = _lib.Ctx(DEV, options={_lib.OPT_NT_BIG: 0}, from_env=False)
    if K >= 2048:                                                        # (12-stage tiles need more / fuller rounds to be routed there by default)
        assert ops.gemm_nt(Ad, Bd, _lib.EPI_BF16, bias=bd, route_only=True) == 5, "expected the 192x256 route (plain epilogue, full rounds)"
    assert ops.gemm_nt(Ad, Bd, _lib.EPI_BF16, bias=bd, route_only=True, ctx=tiled) in (0, 3, 4)     # (any tile form of gemm.hip: same k order)
    big = _lib.Ctx(DEV, options={_lib.OPT_NT_BIG: geo}, from_env=False)      # every epilogue forced onto the kernel (default: plain only)
    guard = 7.0
    nseq = 13
    row2seq = (torch.arange(M) * nseq // M).int().to(DEV)
    sc = torch.tensor([0.0 if i % 5 == 2 else 1.0 / 0.9 for i in range(nseq)]).to(DEV)
    resid = torch.randn(M, N, generator=g).to(DEV)
    aux = rnd_bf(M, N, seed=35).to(DEV)

    def run(epi, ctx, **kw):
        f32 = epi == _lib.EPI_RESID_F32
        out = torch.full((M + 200, N), guard, device=DEV, dtype=torch.float32 if f32 else torch.bfloat16)
        res = ops.gemm_nt(Ad, Bd, epi, out=out[:M], ctx=ctx, **kw)
        assert float((out[M:].float() - guard).abs().max()) == 0.0, "rows past M were written"
        return res
    for p_drop in (0.0, 0.1):
        dk = dict(drop_p=p_drop, drop_seed=91)
        # plain
        if p_drop == 0.0:
            a, b = run(_lib.EPI_BF16, big, bias=bd), run(_lib.EPI_BF16, tiled, bias=bd)
            assert torch.equal(a, b) and relerr(a.float(), ref) < 1e-2
        # GELU pair: u / gelu'(u) as first tensor, and the forward-only form
        for act in (0, 1):
            o2a = torch.empty(M, N, device=DEV, dtype=torch.bfloat16); o2b = torch.empty_like(o2a)
            ua = run(_lib.EPI_BF16_GELU, big, bias=bd, out2=o2a, act=act, **dk)[0]
            ub = run(_lib.EPI_BF16_GELU, tiled, bias=bd, out2=o2b, act=act, **dk)[0]
            assert torch.equal(ua, ub) and torch.equal(o2a, o2b)
            if p_drop == 0.0:
                assert relerr(o2a.float(), F.gelu(ref)) < 1e-2
                if act == 0:
                    assert relerr(ua.float(), ref) < 1e-2
        o2a = torch.empty(M, N, device=DEV, dtype=torch.bfloat16); o2b = torch.empty_like(o2a)
        ops.gemm_nt(Ad, Bd, _lib.EPI_BF16_GELU, bias=bd, out2=o2a, skip_pre=True, ctx=big, **dk)
        ops.gemm_nt(Ad, Bd, _lib.EPI_BF16_GELU, bias=bd, out2=o2b, skip_pre=True, ctx=tiled, **dk)
        assert torch.equal(o2a, o2b)
        # residual + DropPath scale (+ element dropout on the branch output), out of place and in place
        kw = dict(bias=bd, resid=resid, seq_scale=sc, row2seq=row2seq, **dk)
        a, b = run(_lib.EPI_RESID_F32, big, **kw), run(_lib.EPI_RESID_F32, tiled, **kw)
        assert torch.equal(a, b)
        if p_drop == 0.0:
            assert relerr(a, resid.cpu() + sc.cpu()[row2seq.cpu().long()].unsqueeze(1) * ref) < 2e-4
            inplace = resid.clone()
            ops.gemm_nt(Ad, Bd, _lib.EPI_RESID_F32, bias=bd, resid=inplace, seq_scale=sc, row2seq=row2seq, out=inplace, ctx=big)
            assert torch.equal(inplace, a), "in-place residual"
        # GELU' input gradient: saved gelu'(u) multiplied in / derivative evaluated from u
        for act in (0, 1):
            a, b = run(_lib.EPI_DGELU_BF16, big, aux=aux, act=act, **dk), run(_lib.EPI_DGELU_BF16, tiled, aux=aux, act=act, **dk)
            assert torch.equal(a, b)
    x = aux.float().cpu().requires_grad_(True)
    F.gelu(x).sum().backward()
    assert relerr(run(_lib.EPI_DGELU_BF16, big, aux=aux).float(), (A.float() @ B.float().t()) * x.grad) < 1e-2


_TILED_SNIPPET = """
import torch, torch.nn.functional as F
from lafs_cvpr2024_amd import _lib, ops
M, N, K = 128 * 41 + 1, 1536, 384
g = torch.Generator().manual_seed(21)
A = torch.randn(M, K, generator=g).to(torch.bfloat16); B = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16)
aux = torch.randn(M, N, generator=g).to(torch.bfloat16); bias = torch.randn(N, generator=g); resid = torch.randn(M, N, generator=g)
x = aux.float().requires_grad_(True); F.gelu(x).sum().backward()
Ad, Bd, ad, bd = A.cuda(), B.cuda(), aux.cuda(), bias.cuda()
rel = lambda o, r: ((o.float().cpu() - r).abs().max() / r.abs().max()).item()
ref = A.float() @ B.float().t()
for epi, kw in ((_lib.EPI_BF16, {}), (_lib.EPI_BF16_GELU, {}), (_lib.EPI_RESID_F32, dict(resid=resid.cuda())), (_lib.EPI_DGELU_BF16, dict(aux=ad))):
    assert ops.gemm_nt(Ad, Bd, epi, bias=None if epi == _lib.EPI_DGELU_BF16 else bd, route_only=True, **kw) == 0, "LAFS_KRES=0 must route to the tiled kernel"
assert rel(ops.gemm_nt(Ad, Bd, _lib.EPI_BF16, bias=bd), ref + bias) < 1e-2
u, a = ops.gemm_nt(Ad, Bd, _lib.EPI_BF16_GELU, bias=bd)
assert rel(u, ref + bias) < 1e-2 and rel(a, F.gelu(ref + bias)) < 1e-2
assert rel(ops.gemm_nt(Ad, Bd, _lib.EPI_RESID_F32, bias=bd, resid=resid.cuda()), resid + ref + bias) < 2e-4
assert rel(ops.gemm_nt(Ad, Bd, _lib.EPI_DGELU_BF16, aux=ad), ref * x.grad) < 1e-2
print("TILED_OK")
"""


def test_gemm_nt_tiled_kernel_at_the_streaming_shapes():
    """LAFS_KRES=0 (the A/B switch of tools/lab/ab_env.sh, read by _lib.Ctx -> LAFS_OPT_KRES_MASK) sends the K = 384 streaming shapes
    back to the tiled kernel: same four epilogues, same tolerances, in a fresh process (the default context is built once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LAFS_KRES="0", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", _TILED_SNIPPET], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0 and "TILED_OK" in r.stdout, r.stdout + r.stderr


def test_gemm_nt_embed_epilogue():
    nseq, npatch, D = 3, 36, 128
    A, B = rnd_bf(nseq * npatch, 192, seed=1), rnd_bf(D, 192, scale=0.1, seed=2)
    bias, pos = torch.randn(D), torch.randn(npatch + 1, D)
    out = torch.zeros(nseq * (npatch + 1), D, device=DEV)
    ops.gemm_nt(A.to(DEV), B.to(DEV), _lib.EPI_EMBED_F32, bias=bias.to(DEV), pos=pos.to(DEV), npatch=npatch, out=out)
    ref = torch.zeros(nseq, npatch + 1, D)
    ref[:, 1:] = (A.float() @ B.float().t() + bias).view(nseq, npatch, D) + pos[1:]
    assert relerr(out, ref.view(-1, D)) < 1e-4


@pytest.mark.parametrize("M,N1,N2", [(64, 128, 128), (1000, 384, 1152), (777, 200, 72), (640, 1000, 256)])
def test_gemm_tn(M, N1, N2):
    A, B = rnd_bf(M, N1, seed=1), rnd_bf(M, N2, seed=2)
    Cd = torch.zeros(N1, N2, device=DEV)
    ops.gemm_tn_acc(A.to(DEV), B.to(DEV), Cd)
    ref = A.float().t() @ B.float()
    assert relerr(Cd, ref) < 2e-5 * math.sqrt(M)
    cs = torch.zeros(N1, device=DEV)
    ops.gemm_tn_acc(A.to(DEV), B.to(DEV), Cd, splits=3, colsum=cs)      # accumulates; bias gradient rides along
    assert relerr(Cd, 2 * ref) < 2e-5 * math.sqrt(M)
    assert relerr(cs, A.float().sum(0)) < 2e-5 * math.sqrt(M)


@pytest.mark.parametrize("M,N1,N2", [(64, 128, 128), (1000, 384, 1152), (777, 200, 72), (5000, 1536, 384), (4099, 384, 1536),
                                     (3000, 704, 768), (2500, 2112, 768), (130, 384, 384), (9000, 256, 256)])
def test_wgrad_wide_tile(M, N1, N2):
    """lafs_wgrad (slice-partial + fold, no atomics): overwrite, accumulate, bias gradient, ragged token tails and ragged
    output edges, against fp32 math on the same bf16 operands."""
    A, B = rnd_bf(M, N1, seed=1), rnd_bf(M, N2, seed=2)
    ref = A.float().t() @ B.float()
    Cd = torch.full((N1, N2), 7.0, device=DEV)                      # overwrite mode must not read the old contents
    ops.wgrad(A.to(DEV), B.to(DEV), Cd, accumulate=False)
    e0 = relerr(Cd, ref)
    cs = torch.zeros(N1, device=DEV)
    ops.wgrad(A.to(DEV), B.to(DEV), Cd, accumulate=True, colsum=cs)
    e1, e2 = relerr(Cd, 2 * ref), relerr(cs, A.float().sum(0))
    print(f"wgrad {M}x{N1}x{N2}: overwrite {e0:.2e} accumulate {e1:.2e} colsum {e2:.2e}")
    assert e0 < 2e-5 * math.sqrt(M) and e1 < 2e-5 * math.sqrt(M) and e2 < 2e-5 * math.sqrt(M)
    # strided operands (column slices of wider matrices, as the fused qkv gradient uses them)
    Aw, Bw = rnd_bf(M, N1 + 64, seed=3).to(DEV), rnd_bf(M, N2 + 8, seed=4).to(DEV)
    Cs = torch.zeros(N1, N2, device=DEV)
    ops.wgrad(Aw[:, 64:], Bw[:, :N2], Cs, accumulate=False)
    assert relerr(Cs, Aw[:, 64:].float().t() @ Bw[:, :N2].float()) < 2e-5 * math.sqrt(M)


@pytest.mark.parametrize("M,dims", [(5000, [(384, 1536), (1536, 384), (384, 384), (1152, 384)]),
                                    (1217, [(768, 2048), (2048, 768), (2112, 768), (768, 704)]), (300, [(64, 64), (200, 72)])])
def test_wgrad_group(M, dims):
    """The weight gradients of one transformer block as ONE grouped launch: every GEMM equals its own fp32 product; mixed
    overwrite / accumulate flags; bias gradients ride along."""
    probs, refs = [], []
    for i, (n1, n2) in enumerate(dims):
        A, B = rnd_bf(M, n1, seed=10 + i), rnd_bf(M, n2, seed=20 + i)
        acc = bool(i & 1)
        C0 = torch.randn(n1, n2, generator=torch.Generator().manual_seed(30 + i))
        refs.append((A.float().t() @ B.float() + (C0 if acc else 0), A.float().sum(0)))
        probs.append((A.to(DEV), B.to(DEV), C0.to(DEV), acc, torch.zeros(n1, device=DEV)))
    ops.wgrad_group(probs)
    for (A, B, Cd, acc, cs), (ref, csr) in zip(probs, refs):
        assert relerr(Cd, ref) < 2e-5 * math.sqrt(M), (A.shape, B.shape)
        assert relerr(cs, csr) < 2e-5 * math.sqrt(M)


# (rows beyond 4096: a backward wave walks several rows, the next one's operands in flight while it reduces the current one)
@pytest.mark.parametrize("rows,D", [(50, 64), (1000, 384), (333, 192), (64, 768), (7, 2048), (9001, 384), (5003, 768), (8200, 192),
                                    (4100, 1024)])
def test_layernorm_fwd_bwd(rows, D):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(rows, D, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(D, generator=g), torch.randn(D, generator=g)
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (D,), gr, br, 1e-6)
    y, yf, stats = ops.layernorm_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-6, want_f32=True)
    assert relerr(yf, yr) < 1e-5 and relerr(y.float(), yr) < 1e-2
    dy = rnd_bf(rows, D, seed=9)
    yr.backward(dy.float())
    g0 = torch.randn(rows, D, generator=g)
    nseq = 5
    row2seq = (torch.arange(rows) * nseq // rows).int()
    sc = torch.tensor([1.0, 0.0, 2.0, 1.0, 0.5])
    g_io = g0.clone().to(DEV)
    dgam, dbet = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    gb = torch.empty(rows, D, device=DEV, dtype=bf16)
    ops.layernorm_bwd(dy.to(DEV), x.to(DEV), stats, gamma.to(DEV), g_io, dgam, dbet, accumulate=True, gb_out=gb,
                      seq_scale=sc.to(DEV), row2seq=row2seq.to(DEV))
    assert relerr(g_io, g0 + xr.grad) < 1e-5
    assert relerr(dgam, gr.grad) < 1e-4 and relerr(dbet, br.grad) < 1e-4
    assert relerr(gb.float(), sc[row2seq.long()].unsqueeze(1) * (g0 + xr.grad)) < 1e-2
    # fp32 upstream gradient (the final norm's), written instead of accumulated
    g2 = torch.full((rows, D), 5.0, device=DEV)
    dgam.zero_(); dbet.zero_()
    ops.layernorm_bwd(dy.float().to(DEV), x.to(DEV), stats, gamma.to(DEV), g2, dgam, dbet, accumulate=False)
    assert relerr(g2, xr.grad) < 1e-5
    assert relerr(dgam, gr.grad) < 1e-4 and relerr(dbet, br.grad) < 1e-4


def _attn_ref(qkv, cu, heads, scale):
    inner = heads * 64
    outs = []
    for s in range(len(cu) - 1):
        x = qkv[cu[s]:cu[s + 1]]
        n = x.shape[0]
        q, k, v = (x[:, i * inner:(i + 1) * inner].view(n, heads, 64).transpose(0, 1) for i in range(3))
        a = (q @ k.transpose(-1, -2) * scale).softmax(-1)
        outs.append((a @ v).transpose(0, 1).reshape(n, inner))
    return torch.cat(outs)


@pytest.mark.parametrize("lens,heads", [([197, 197], 2), ([37] * 5, 3), ([16, 1, 37, 48, 33], 1), ([100, 77], 2),
                                        ([197, 150], 6), ([256, 200], 1), ([160, 130, 9], 3), ([197] * 90, 3)])
def test_attention_fwd_bwd(lens, heads):
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + n)
    T, inner = cu[-1], heads * 64
    qkv = rnd_bf(T, 3 * inner, seed=11)
    scale = 64 ** -0.5
    xr = qkv.float().requires_grad_(True)
    ref = _attn_ref(xr, cu, heads, scale)
    cud = torch.tensor(cu, dtype=torch.int32, device=DEV)
    out, lse = ops.attention_fwd(qkv.to(DEV), cud, max(lens), heads, scale)
    assert relerr(out.float(), ref) < 1.5e-2
    dout = rnd_bf(T, inner, seed=12)
    ref.backward(dout.float())
    dqkv = ops.attention_bwd(qkv.to(DEV), out, dout.to(DEV), lse, cud, max(lens), heads, scale)
    for i, name in enumerate("qkv"):
        e = relerr(dqkv[:, i * inner:(i + 1) * inner].float(), xr.grad[:, i * inner:(i + 1) * inner])
        assert e < 3e-2, f"d{name}: {e}"


@pytest.mark.parametrize("k,stride,H", [(3, 1, 14), (3, 2, 56), (5, 1, 7), (5, 2, 7), (5, 2, 28), (3, 2, 9)])
def test_depthwise_conv_nchw_forward_and_gradients(k, stride, H):
    """DepthwiseConv2d (trainable landmark branch): HIP forward / data gradient / weight gradient vs torch's grouped
    convolution in fp32 (same arithmetic, different summation order)."""
    import torch.nn.functional as F
    from lafs_cvpr2024_amd.face_pre_pro.mobilenet import DepthwiseConv2d
    torch.manual_seed(k * 10 + stride)
    N, C = 3, 24
    conv = DepthwiseConv2d(C, k, stride).to("cuda")
    x = torch.randn(N, C, H, H, device="cuda", requires_grad=True)
    y = conv(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    xr = x.detach().clone().requires_grad_(True)
    wr = conv.weight.detach().clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, (k - 1) // 2, 1, C)
    yr.backward(gy)
    assert y.shape == yr.shape
    torch.testing.assert_close(y, yr, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(x.grad, xr.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(conv.weight.grad, wr.grad, rtol=1e-4, atol=1e-4)


def test_per_tensor_grad_norms_need_no_zeroing_and_repeat_bitwise():
    """utils.clip_gradients' per-tensor norms (reference utils.py:132-141) over a flat arena of ragged tensors: equal to the
    per-tensor fp64 sums, independent of what the output and scratch buffers held before, and bitwise equal from call to call."""
    from lafs_cvpr2024_amd.ops import _p, call
    g = torch.Generator().manual_seed(3)
    chunks = [1, 13, 12, 1, 40, 3, 1, 700, 2]                              # chunks per tensor; one block of the old kernel saw many
    n_seg, n_chunks = len(chunks), sum(chunks)
    grad = torch.randn(n_chunks * _lib.CHUNK, generator=g).to(DEV)
    chunk_seg = torch.tensor(sum(([i] * c for i, c in enumerate(chunks)), []), dtype=torch.int32, device=DEV)
    hyper = torch.zeros(_lib.HP_COUNT, device=DEV); hyper[_lib.HP_GRAD_SCALE] = 0.5
    outs = []
    for junk in (float("nan"), -7.0):
        seg = torch.full((n_seg,), junk, device=DEV); scratch = torch.full((n_chunks,), junk, device=DEV)
        call("lafs_grad_sumsq", _p(grad), _p(chunk_seg), n_chunks, n_seg, _p(hyper), _p(scratch), _p(seg))
        outs.append(seg.cpu())
    assert torch.equal(outs[0], outs[1])
    ref = torch.stack([t.double().pow(2).sum() * 0.25 for t in grad.cpu().split([c * _lib.CHUNK for c in chunks])])
    err = ((outs[0].double() - ref).abs() / ref).max().item()
    print(f"per-tensor sumsq max rel err {err:.2e}")
    assert err < 1e-5


def test_invalid_arguments_fail_loudly():
    """Unsupported shapes are rejected by the C entry points (negative return code -> LafsHipError with the reason), never
    silently mis-computed or routed to another implementation."""
    from lafs_cvpr2024_amd import _lib
    from lafs_cvpr2024_amd.ops import _p, call
    A = torch.zeros(64, 40, device=DEV, dtype=torch.bfloat16)            # K = 40 is not a multiple of 32
    B = torch.zeros(128, 40, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(_lib.LafsHipError, match="multiple of 32"):
        ops.gemm_nt(A, B)
    qkv = torch.zeros(300, 192, device=DEV, dtype=torch.bfloat16)
    cu = torch.tensor([0, 300], dtype=torch.int32, device=DEV)
    with pytest.raises(_lib.LafsHipError, match="1..256"):
        ops.attention_fwd(qkv, cu, 300, 1, 0.125)                         # sequences longer than 256 tokens
    img = torch.zeros(1, 3, 112, 112, device=DEV); th = torch.zeros(1, 30, 2, device=DEV); out = torch.zeros(1, 3, 48, 48, device=DEV)
    with pytest.raises(_lib.LafsHipError, match="perfect square"):
        call("lafs_patch_gather_fwd", _p(img), _p(th), 1, 112, 30, _p(out))
    x = torch.zeros(4, 100, device=DEV)
    with pytest.raises(_lib.LafsHipError):
        ops.gemm_nt(x, B)                                                 # fp32 operand where bf16 is required
    with pytest.raises(_lib.LafsHipError, match="drop_p"):
        ops.gemm_nt(torch.zeros(64, 64, device=DEV, dtype=torch.bfloat16), torch.zeros(128, 64, device=DEV, dtype=torch.bfloat16),
                    _lib.EPI_RESID_F32, resid=torch.zeros(64, 128, device=DEV), drop_p=1.5)


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("act_name,H", [("relu", 14), ("hswish", 7), ("none", 4), ("hswish", 56)])
def test_fused_batchnorm_activation_matches_torch(training, act_name, H, monkeypatch):
    """bn_act (fused BatchNorm2d + activation, fp32 NCHW) vs nn.BatchNorm2d followed by the activation module: output,
    running statistics, and the three gradients, in training and eval mode (7x7 planes exercise the non-float4 path)."""
    import copy
    from lafs_cvpr2024_amd.face_pre_pro.mobilenet import bn_act
    monkeypatch.setenv("LAFS_BN_FUSED", "1")               # opt-in path (the default keeps MIOpen's BatchNorm: it is faster)
    torch.manual_seed(H)
    N, C = 6, 24
    bn = torch.nn.BatchNorm2d(C).to("cuda")
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    ref_bn = copy.deepcopy(bn)
    act = {"relu": torch.nn.ReLU(), "hswish": torch.nn.Hardswish(), "none": None}[act_name]
    bn.train(training); ref_bn.train(training)
    x = (torch.randn(N, C, H, H, device="cuda") * 2 + 0.5).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    y = bn_act(x, bn, act)
    yr = ref_bn(xr)
    yr = act(yr) if act is not None else yr
    torch.testing.assert_close(y, yr, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(bn.running_mean, ref_bn.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var, ref_bn.running_var, rtol=1e-4, atol=1e-5)
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked)
    gy = torch.randn_like(y)
    y.backward(gy); yr.backward(gy)
    scale = float(xr.grad.abs().max())
    assert float((x.grad - xr.grad).abs().max()) < 2e-4 * max(scale, 1.0)
    torch.testing.assert_close(bn.weight.grad, ref_bn.weight.grad, rtol=1e-3, atol=2e-3)
    torch.testing.assert_close(bn.bias.grad, ref_bn.bias.grad, rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("K,ld,ncrops,B", [(1003, 1004, 10, 6), (4096, 4096, 4, 5), (2, 4, 2, 3), (100000, 100096, 3, 2)])
@pytest.mark.parametrize("grad_bf16", [True, False])
def test_dino_loss_kernels_against_torch(K, ld, ncrops, B, grad_bf16):
    """Fused sharpen / center / softmax / cross-entropy (reference lafs_train.py:643-667) on padded logits with a ragged class
    count: loss, dL/dstudent (autograd of the plain formula) and the column sums / center EMA of update_center (:669-679)."""
    g = torch.Generator().manual_seed(K + ncrops)
    s = torch.zeros(ncrops * B, ld); t = torch.zeros(2 * B, ld)
    s[:, :K] = torch.randn(ncrops * B, K, generator=g) * 3; t[:, :K] = torch.randn(2 * B, K, generator=g) * 2
    center = torch.randn(K, generator=g) * 0.1
    ts, tt = 0.1, 0.04
    sr = s[:, :K].clone().double().requires_grad_(True)
    q = F.softmax((t[:, :K].double() - center.double()) / tt, dim=-1).chunk(2)
    lp = F.log_softmax(sr / ts, dim=-1).chunk(ncrops)
    total, n = 0.0, 0
    for iq, qq in enumerate(q):
        for v in range(ncrops):
            if v == iq:
                continue
            total = total + torch.sum(-qq * lp[v], dim=-1).mean(); n += 1
    ref = total / n
    ref.backward()
    cpad = torch.zeros(ld); cpad[:K] = center
    loss, grad = ops.dino_loss_fwd_bwd(s.to(DEV), t.to(DEV), cpad.to(DEV), ncrops, ts, tt, K=K, grad_bf16=grad_bf16)
    assert abs(float(loss) - float(ref.detach())) < 2e-5 * abs(float(ref.detach())) + 1e-6
    gr = grad[:, :K].float().cpu()
    tol = (8e-3 if grad_bf16 else 2e-5) * float(sr.grad.abs().max())
    assert float((gr - sr.grad.float()).abs().max()) < tol
    assert float(grad[:, K:].float().abs().max()) == 0.0 if ld > K else True
    colsum = torch.zeros(ld, device=DEV)
    ops.call("lafs_colsum_f32", ops._p(t.to(DEV)), ld, 2 * B, K, ops._p(colsum))
    assert torch.allclose(colsum[:K].cpu(), t[:, :K].sum(0), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("K,Kpad,D,with_t", [(1000, 1024, 256, True), (77, 128, 256, True), (130, 192, 64, False), (65, 128, 512, False)])
def test_weightnorm_forward_rows_and_transposed_copy(K, Kpad, D, with_t):
    """nn.utils.weight_norm(dim=0) of DINOHead.last_layer (reference vision_transformer.py:282-285): w = g v / ||v|| per class row,
    the zero-filled pad rows, 1/||v|| and the transposed bf16 copy the head backward reads."""
    g_ = torch.Generator().manual_seed(K)
    v = torch.randn(K, D, generator=g_).to(DEV); g = (torch.rand(K, generator=g_) + 0.5).to(DEV)
    w = torch.full((Kpad, D), 7.0, device=DEV, dtype=bf16)
    wt = torch.full((D, Kpad), 7.0, device=DEV, dtype=bf16) if with_t else None
    inv = torch.empty(K, device=DEV)
    ops.call("lafs_weightnorm_fwd", ops._p(v), ops._p(g), K, Kpad, D, ops._p(w), ops._p(wt), Kpad, ops._p(inv))
    ref = (g[:, None] * v / v.norm(dim=1, keepdim=True))
    assert torch.allclose(inv, 1.0 / v.norm(dim=1), rtol=1e-5)
    assert float((w[:K].float() - ref).abs().max()) < 8e-3 * float(ref.abs().max())
    assert float(w[K:].float().abs().max()) == 0.0
    if with_t:
        assert torch.equal(wt, w.t().contiguous())


def test_transposed_weight_shadows_table_kernel():
    """All W^T bf16 shadows in one launch (lafs_transpose_cast_table): matrices whose sides are / are not multiples of the 64x64
    tile and of the 8-row store pieces, destination offsets aligned and not."""
    shapes = [(1152, 384), (384, 192), (100, 37), (8, 2048), (257, 65), (64, 64)]
    g_ = torch.Generator().manual_seed(3)
    mats = [torch.randn(r, c, generator=g_) for r, c in shapes]
    master = torch.cat([torch.zeros(5)] + [m.reshape(-1) for m in mats]).to(DEV)       # (matrix offsets not 16-byte aligned)
    rows_, starts, n, off, toff = [], [0], 0, 5, 3
    toffs = []
    for r, c in shapes:
        rows_ += [off, r, c, toff]; toffs.append(toff)
        off += r * c; toff += r * c + (5 if r == 100 else 0)
        n += ((r + 63) // 64) * ((c + 63) // 64); starts.append(n)
    shadow = torch.full((toff + 8,), 9.0, device=DEV, dtype=bf16)
    table = torch.tensor(rows_, dtype=torch.int64, device=DEV); st = torch.tensor(starts, dtype=torch.int32, device=DEV)
    ops.call("lafs_transpose_cast_table", ops._p(master), ops._p(shadow), ops._p(table), ops._p(st), len(shapes), n)
    for m, (r, c), to in zip(mats, shapes, toffs):
        got = shadow[to:to + r * c].view(c, r).float().cpu()
        assert torch.equal(got, m.t().to(bf16).float())
    assert float(shadow[:3].float().min()) == 9.0 and float(shadow[toff:].float().min()) == 9.0      # nothing written outside


def test_dropout_seed_from_a_device_step_counter_and_row_offsets():
    """Element dropout for graph-captured steps (lafs_hip.h: drop_step / drop_row0): the seed of a launch is drop_seed + 7919 * step
    with `step` read from DEVICE memory when the kernel runs, and a launch over rows [r0, r0 + R) of a batch applies rows r0.. of the
    batch's mask.  Every site -- residual epilogue, GELU epilogue, GELU' epilogue, LayerNorm-backward / cast gradients, in-place
    embedding dropout -- must reproduce the mask lafs_debug_dropout_mask exports for the host-side seed' = seed + 7919 * step."""
    from lafs_cvpr2024_amd import _lib
    from lafs_cvpr2024_amd.ops import _p, call
    torch.manual_seed(5)
    M, N, K, p, seed, stepv, r0 = 320, 256, 128, 0.25, 0x1234, 37.0, 96
    step = torch.tensor([stepv], device=DEV)
    eff = (seed + 7919 * int(stepv)) & 0xFFFFFFFF
    mask = ops.dropout_mask(M, N, p, eff, DEV)
    assert abs(float((mask > 0).float().mean()) - (1 - p)) < 0.02
    assert not torch.equal(mask, ops.dropout_mask(M, N, p, seed, DEV))            # the step changes the mask
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV) * 0.1).to(torch.bfloat16)
    resid = torch.randn(M, N, device=DEV)
    plain = ops.gemm_nt(A, W, _lib.EPI_RESID_F32, resid=resid) - resid
    # residual epilogue: the whole batch, then a row sub-range with its offset
    full = ops.gemm_nt(A, W, _lib.EPI_RESID_F32, resid=resid, drop_p=p, drop_seed=seed, drop_step=step) - resid
    torch.testing.assert_close(full, plain * mask, rtol=1e-5, atol=1e-5)
    part = ops.gemm_nt(A[r0:], W, _lib.EPI_RESID_F32, resid=resid[r0:], drop_p=p, drop_seed=seed, drop_step=step, drop_row0=r0) - resid[r0:]
    torch.testing.assert_close(part, full[r0:], rtol=0, atol=0)
    wrong = ops.gemm_nt(A[r0:], W, _lib.EPI_RESID_F32, resid=resid[r0:], drop_p=p, drop_seed=seed, drop_step=step) - resid[r0:]
    assert not torch.equal(wrong, full[r0:])                                      # without the offset: launch-relative rows
    # the counter is read when the kernel runs: same arguments, new value -> new mask
    step.fill_(stepv + 1)
    full2 = ops.gemm_nt(A, W, _lib.EPI_RESID_F32, resid=resid, drop_p=p, drop_seed=seed, drop_step=step) - resid
    torch.testing.assert_close(full2, plain * ops.dropout_mask(M, N, p, (seed + 7919 * int(stepv + 1)) & 0xFFFFFFFF, DEV), rtol=1e-5, atol=1e-5)
    step.fill_(stepv)
    # GELU epilogue (mask on GELU(u) only) and its backward
    _, a_plain = ops.gemm_nt(A, W, _lib.EPI_BF16_GELU)
    u, a_drop = ops.gemm_nt(A[r0:], W, _lib.EPI_BF16_GELU, drop_p=p, drop_seed=seed, drop_step=step, drop_row0=r0)
    torch.testing.assert_close(a_drop.float(), (a_plain.float() * mask)[r0:].to(torch.bfloat16).float(), rtol=1e-2, atol=1e-3)
    dy = torch.randn(M - r0, K, device=DEV).to(torch.bfloat16)
    d_plain = ops.gemm_nt(dy, W, _lib.EPI_DGELU_BF16, aux=u)
    d_drop = ops.gemm_nt(dy, W, _lib.EPI_DGELU_BF16, aux=u, drop_p=p, drop_seed=seed, drop_step=step, drop_row0=r0)
    torch.testing.assert_close(d_drop.float(), (d_plain.float() * mask[r0:]).to(torch.bfloat16).float(), rtol=1e-2, atol=1e-3)
    # gradient casts
    g = torch.randn(M, N, device=DEV)
    gb = ops.scale_cast_bf16(g[r0:], drop_p=p, drop_seed=seed, drop_step=step, drop_row0=r0)
    torch.testing.assert_close(gb.float(), (g * mask)[r0:].to(torch.bfloat16).float(), rtol=0, atol=0)
    x = torch.randn(M, N, device=DEV)
    x2 = x.clone()
    call("lafs_dropout_f32", _p(x2), N, M, N, p, seed, _p(step))
    torch.testing.assert_close(x2, x * mask, rtol=0, atol=0)


@pytest.mark.parametrize("name,M,dims", [("ViT-S block", 44160, [(384, 1536), (1536, 384), (384, 384), (1152, 384)]),
                                         ("Part-fViT block, one row chain", 25216, [(768, 2048), (2048, 768), (768, 704), (2112, 768)])])
def test_wgrad_group_under_the_engines_workgroup_caps_at_block_shapes(name, M, dims):
    """The grouped weight gradient as the engines launch it beside the dgrad chain (max_workgroups 200; 0 = the whole chip; other caps
    change tile shape and slice count), written and accumulated, with the bias-gradient column sums shared over tiles and wave
    columns: against fp32 torch at the full block shapes."""
    from lafs_cvpr2024_amd import ops
    torch.manual_seed(0)
    pairs = [(torch.randn(M, a, device="cuda").to(torch.bfloat16), torch.randn(M, b, device="cuda").to(torch.bfloat16)) for a, b in dims]
    refs = [x.float().t() @ y.float() for x, y in pairs]
    cref = [x.float().sum(0) for x, _ in pairs]
    for cap in (0, 200, 96):
        for acc in (False, True):
            Cs = [torch.full((a, b), 0.5 if acc else 7.0, device="cuda") for a, b in dims]
            cs = [torch.zeros(a, device="cuda") for a, _ in dims]
            ops.wgrad_group([(x, y, c, acc, s) for (x, y), c, s in zip(pairs, Cs, cs)], max_workgroups=cap)
            for c, r, s, sr in zip(Cs, refs, cs, cref):
                assert float(((c - (0.5 if acc else 0.0)) - r).norm() / r.norm()) < 1e-4, (name, cap, acc, tuple(c.shape))
                assert float((s - sr).norm() / sr.norm()) < 1e-4, (name, cap, acc, "colsum")


@pytest.mark.parametrize("ncrops,B,K", [(4, 4, 1000), (10, 3, 8192), (10, 64, 100000)])
def test_fused_head_and_dino_loss_equal_the_unfused_kernels(ncrops, B, K):
    """lafs_dino_head_loss (the last layer of both DINO heads + DINOLoss.forward + the centre's row sums with the logits formed twice
    on the MFMA and never stored; vision_transformer.py:295-301, lafs_train.py:643-679) against the unfused kernels it replaces in
    the training step -- lafs_gemm_nt(F32) -> lafs_dino_loss_fwd_bwd + lafs_colsum_f32, themselves pinned by F4 -- on the same
    operands: loss to 1e-6 relative, dL/dlogits equal up to single bf16 roundings (the exponentials are evaluated in another order),
    pad columns exactly zero, centre sums to fp32 round-off; class counts that are no multiple of the 64-class block, a temperature
    read from device memory; and twice in a row bit for bit (no atomics anywhere)."""
    g = torch.Generator().manual_seed(40 + ncrops)
    Kpad = (K + 127) // 128 * 128
    nrm = lambda t: torch.nn.functional.normalize(t, dim=1)
    zs, zt = nrm(torch.randn(ncrops * B, 256, generator=g)).to(bf16).to(DEV), nrm(torch.randn(2 * B, 256, generator=g)).to(bf16).to(DEV)
    ws, wt = torch.zeros(Kpad, 256, dtype=bf16), torch.zeros(Kpad, 256, dtype=bf16)
    ws[:K] = nrm(torch.randn(K, 256, generator=g)).to(bf16); wt[:K] = nrm(torch.randn(K, 256, generator=g)).to(bf16)
    ws, wt = ws.to(DEV), wt.to(DEV)
    center = (0.05 * torch.randn(K, generator=g)).to(DEV)
    temps = torch.tensor([0.1, 0.055], device=DEV)
    # unfused
    ls = ops.gemm_nt(zs, ws, _lib.EPI_F32, n_cols=Kpad); lt = ops.gemm_nt(zt, wt, _lib.EPI_F32, n_cols=Kpad)
    loss_ref, grad_ref = ops.dino_loss_fwd_bwd(ls, lt, center, ncrops, 0.1, 0.04, K=K, dev_temps=temps)
    col_ref = torch.empty(K, device=DEV)
    call("lafs_colsum_f32", _p(lt), Kpad, 2 * B, K, _p(col_ref))
    # fused
    guard = 3.0
    grad = torch.full((ncrops * B + 8, Kpad), guard, device=DEV, dtype=bf16)
    col = torch.full((K + 8,), guard, device=DEV)
    loss, _ = ops.dino_head_loss(zs, zt, ws, wt, center, ncrops, K, 0.1, 0.04, grad=grad[:ncrops * B], colsum=col, dev_temps=temps)
    assert abs(float(loss) - float(loss_ref)) <= 2e-6 * abs(float(loss_ref)), (float(loss), float(loss_ref))
    assert float(loss) == pytest.approx(math.log(K), rel=0.2)
    gf, gr = grad[:ncrops * B].float(), grad_ref.float()
    scale = float(gr.abs().max())
    assert float((gf - gr).abs().max()) <= 2.0 ** -7 * scale and relerr(gf, gr) < 1e-2
    assert float((gf != gr).float().mean()) < 0.05, "more than single roundings apart"
    assert float(gf[:, K:].abs().max()) == 0.0 if Kpad > K else True
    assert torch.all(grad[ncrops * B:] == guard) and torch.all(col[K:] == guard), "wrote past the end"
    torch.testing.assert_close(col[:K], col_ref, rtol=1e-5, atol=1e-5)
    grad2 = torch.empty_like(grad[:ncrops * B]); col2 = torch.empty(K, device=DEV)
    loss2, _ = ops.dino_head_loss(zs, zt, ws, wt, center, ncrops, K, 0.1, 0.04, grad=grad2, colsum=col2, dev_temps=temps)
    assert torch.equal(loss2, loss) and torch.equal(grad2, grad[:ncrops * B]) and torch.equal(col2, col[:K])
    # host temperatures (no device override) give the same as the device pair holding the same values
    temps2 = torch.tensor([0.1, 0.04], device=DEV)
    la, ga = ops.dino_head_loss(zs, zt, ws, wt, center, ncrops, K, 0.1, 0.04, grad=torch.empty_like(grad2))
    lb, gb = ops.dino_head_loss(zs, zt, ws, wt, center, ncrops, K, 0.1, 0.04, grad=torch.empty_like(grad2), dev_temps=temps2)
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(lb)) and relerr(ga.float(), gb.float()) < 1e-2


@pytest.mark.parametrize("M,H", [(128 * 19, 1536), (128 * 17 + 37, 1536), (128 * 16 + 16 * 5 + 3, 768), (25216, 1536), (100, 128)])
def test_fused_mlp_equals_the_two_launches_bit_for_bit(M, H):
    """lafs_mlp_fused (csrc/mlp_fused.hip; Mlp.forward + the residual of Block.forward, vision_transformer.py:59-65,112, and its
    input-gradient chain) against the two lafs_gemm_nt launches each mode replaces: same MFMA, same operand roles, same ascending k
    order, same bias / GELU / residual arithmetic -> 0 differing bits in every output (fp32 result, saved gelu'(u) / gelu(u), du,
    dX); against fp32 torch within the bf16 tolerances; ragged last unit (rows past M untouched), partly and wholly idle waves."""
    D = 384
    g = torch.Generator().manual_seed(71)
    X = rnd_bf(M, D, seed=72).to(DEV)
    W1, W2 = rnd_bf(H, D, scale=0.05, seed=73).to(DEV), rnd_bf(D, H, scale=0.03, seed=74).to(DEV)
    b1, b2 = (torch.randn(H, generator=g) * 0.1).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV)
    resid = torch.randn(M, D, generator=g).to(DEV)
    nseq = 11
    row2seq = (torch.arange(M) * nseq // M).int().to(DEV)
    sc = torch.tensor([0.0 if i % 4 == 1 else 1.0 / 0.9 for i in range(nseq)]).to(DEV)
    guard = 5.0
    # GEMM 1 of either direction accumulates on top of its bias like the K-resident kernel (gemm_kres.hip, rows >= 2048); the tiled
    # kernel adds the bias to the finished sum, which rounds differently: below 2048 rows the comparison is by tolerance
    exact = M >= 2048
    if exact:
        assert ops.gemm_nt(X, W1, _lib.EPI_BF16_GELU, bias=b1, route_only=True) == 1
    same = (lambda a, b: torch.equal(a, b)) if exact else (lambda a, b: relerr(a.float(), b.float()) < 8e-3)
    # ---- two launches (the kernels of the K-resident / tiled routes)
    a_t = ops.gemm_nt(X, W1, _lib.EPI_BF16_GELU, bias=b1, skip_pre=True)[1]                        # teacher form: gelu(u) only
    y_t = ops.gemm_nt(a_t, W2, _lib.EPI_RESID_F32, bias=b2, resid=resid, seq_scale=sc, row2seq=row2seq)
    gs, a_s = ops.gemm_nt(X, W1, _lib.EPI_BF16_GELU, bias=b1, act=1)                               # saving form: gelu'(u), gelu(u)
    y_s = ops.gemm_nt(a_s, W2, _lib.EPI_RESID_F32, bias=b2, resid=resid, seq_scale=sc, row2seq=row2seq)
    assert torch.equal(a_t, a_s)
    # ---- fused forward, both modes
    out = torch.full((M + 130, D), guard, device=DEV)
    y = ops.mlp_fused(X, W1, W2, _lib.MLP_FWD, bias_a=b1, bias_b=b2, resid=resid, seq_scale=sc, row2seq=row2seq, out=out[:M])[0]
    assert float((out[M:] - guard).abs().max()) == 0.0, "rows past M were written"
    assert same(y, y_t), f"forward-only: {int((y != y_t).sum())} differing values, max {float((y - y_t).abs().max()):.3e}"
    sg = torch.full((M + 130, H), guard, device=DEV, dtype=bf16); sa = torch.full((M + 130, H), guard, device=DEV, dtype=bf16)
    y2, g2, a2 = ops.mlp_fused(X, W1, W2, _lib.MLP_FWD_SAVE, bias_a=b1, bias_b=b2, resid=resid, seq_scale=sc, row2seq=row2seq,
                               save_grad=sg[:M], save_act=sa[:M])
    assert float((sg[M:].float() - guard).abs().max()) == 0.0 and float((sa[M:].float() - guard).abs().max()) == 0.0
    assert same(y2, y_s) and same(g2, gs) and same(a2, a_s)
    # in place on the residual stream (resid aliases out: every lane reads its pieces before it writes them)
    r2 = resid.clone()
    ops.mlp_fused(X, W1, W2, _lib.MLP_FWD, bias_a=b1, bias_b=b2, resid=r2, seq_scale=sc, row2seq=row2seq, out=r2)
    assert same(r2, y_t)
    # against fp32 math
    u = X.float() @ W1.float().t() + b1
    ref = resid + sc[row2seq.long()].unsqueeze(1) * (F.gelu(u).to(bf16).float() @ W2.float().t() + b2)
    assert relerr(y, ref) < 2e-3
    # ---- backward: du = (dY W2) * gelu'(u), dX = du W1, on the transposed shadows
    dY = rnd_bf(M, D, seed=75).to(DEV)
    W2t, W1t = W2.t().contiguous(), W1.t().contiguous()                                            # [H, D], [D, H]
    du_ref = ops.gemm_nt(dY, W2t, _lib.EPI_DGELU_BF16, aux=gs, act=1)
    dx_ref = ops.gemm_nt(du_ref, W1t, _lib.EPI_BF16)
    dub = torch.full((M + 130, H), guard, device=DEV, dtype=bf16); dxb = torch.full((M + 130, D), guard, device=DEV, dtype=bf16)
    dx, _, du = ops.mlp_fused(dY, W2t, W1t, _lib.MLP_BWD, out=dxb[:M], save_grad=gs, save_act=dub[:M])
    assert float((dub[M:].float() - guard).abs().max()) == 0.0 and float((dxb[M:].float() - guard).abs().max()) == 0.0
    assert same(du, du_ref), f"du: {int((du != du_ref).sum())} differing values"
    assert same(dx, dx_ref), f"dX: {int((dx != dx_ref).sum())} differing values"
    dxf = ((dY.float() @ W2.float()) * gs.float()).to(bf16).float() @ W1.float()
    assert relerr(dx.float(), dxf) < 1e-2


@pytest.mark.parametrize("M", [128 * 40, 128 * 33 + 50, 25216])
def test_fused_mlp_layernorm_prologue_equals_the_layernorm_launch_bit_for_bit(M):
    """LayerNorm 2 (vision_transformer.py:112 `self.norm2`, eps 1e-6) computed inside lafs_mlp_fused from the fp32 residual stream:
    the operand it forms, the (mean, rstd) statistics and every output of the fused kernel equal lafs_layernorm_fwd followed by the
    fused kernel on its output, bit for bit (>= 4096 rows: the two-rows-per-wave LayerNorm kernel, whose row arithmetic the prologue
    repeats), forward-only and saving form, ragged last unit."""
    D, H = 384, 1536
    g = torch.Generator().manual_seed(81)
    x = (torch.randn(M, D, generator=g) * 1.7 + 0.3).to(DEV)
    gam, bet = (1.0 + 0.2 * torch.randn(D, generator=g)).to(DEV), (0.1 * torch.randn(D, generator=g)).to(DEV)
    W1, W2 = rnd_bf(H, D, scale=0.05, seed=83).to(DEV), rnd_bf(D, H, scale=0.03, seed=84).to(DEV)
    b1, b2 = (torch.randn(H, generator=g) * 0.1).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV)
    nseq = 9
    row2seq = (torch.arange(M) * nseq // M).int().to(DEV)
    sc = torch.tensor([0.0 if i % 4 == 1 else 1.0 / 0.9 for i in range(nseq)]).to(DEV)
    h = torch.empty(M, D, device=DEV, dtype=bf16); st = torch.empty(M, 2, device=DEV)
    call("lafs_layernorm_fwd", _p(x), D, _p(gam), _p(bet), 1e-6, _p(h), D, None, 0, _p(st), M, D)
    kw = dict(bias_a=b1, bias_b=b2, resid=x, seq_scale=sc, row2seq=row2seq)
    y_ref = ops.mlp_fused(h, W1, W2, _lib.MLP_FWD, **kw)[0]
    y = ops.mlp_fused(None, W1, W2, _lib.MLP_FWD, ln=(gam, bet, 1e-6), **kw)[0]
    assert torch.equal(y, y_ref), f"{int((y != y_ref).sum())} differing values, max {float((y - y_ref).abs().max()):.3e}"
    guard = 3.0
    h2 = torch.full((M + 64, D), guard, device=DEV, dtype=bf16); st2 = torch.full((M + 64, 2), guard, device=DEV)
    y2, g2, a2 = ops.mlp_fused(None, W1, W2, _lib.MLP_FWD_SAVE, ln=(gam, bet, 1e-6), ln_stats=st2[:M], ln_out=h2[:M], **kw)
    yr, gr, ar = ops.mlp_fused(h, W1, W2, _lib.MLP_FWD_SAVE, **kw)
    assert float((h2[M:].float() - guard).abs().max()) == 0.0 and float((st2[M:] - guard).abs().max()) == 0.0, "rows past M were written"
    assert torch.equal(h2[:M], h), f"operand: {int((h2[:M] != h).sum())} differing values"
    assert torch.equal(st2[:M], st)
    assert torch.equal(y2, yr) and torch.equal(g2, gr) and torch.equal(a2, ar)
    ref = F.layer_norm(x.float().cpu(), (D,), gam.cpu(), bet.cpu(), 1e-6)
    assert relerr(h.float(), ref) < 8e-3


@pytest.mark.parametrize("M", [128 * 33, 128 * 33 + 50])
@pytest.mark.parametrize("mode", ["fwd", "save"])
def test_fused_mlp_writes_the_next_blocks_layernorm(M, mode):
    """LayerNorm 1 of block l + 1 (vision_transformer.py:110 `self.norm1`, eps 1e-6) in the fused MLP's final epilogue, from the rows
    the launch holds in registers: the residual stream `out` is bit-identical to the launch without it, and the bf16 operand and the
    (mean, rstd) statistics are bit-identical to lafs_layernorm_fwd on `out` (>= 4096 rows: the two-rows-per-wave kernel, whose
    summation order the epilogue repeats); against torch's fp32 layer_norm; rows past M are not written."""
    D, H = 384, 1536
    g = torch.Generator().manual_seed(101)
    x = (torch.randn(M, D, generator=g) * 1.7 + 0.3).to(DEV)
    gam, bet = (1.0 + 0.2 * torch.randn(D, generator=g)).to(DEV), (0.1 * torch.randn(D, generator=g)).to(DEV)
    ngam, nbet = (1.0 + 0.2 * torch.randn(D, generator=g)).to(DEV), (0.1 * torch.randn(D, generator=g)).to(DEV)
    W1, W2 = rnd_bf(H, D, scale=0.05, seed=103).to(DEV), rnd_bf(D, H, scale=0.03, seed=104).to(DEV)
    b1, b2 = (torch.randn(H, generator=g) * 0.1).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV)
    nseq = 9
    row2seq = (torch.arange(M) * nseq // M).int().to(DEV)
    sc = torch.tensor([0.0 if i % 4 == 1 else 1.0 / 0.9 for i in range(nseq)]).to(DEV)
    md = _lib.MLP_FWD if mode == "fwd" else _lib.MLP_FWD_SAVE
    kw = dict(bias_a=b1, bias_b=b2, resid=x, seq_scale=sc, row2seq=row2seq, ln=(gam, bet, 1e-6))
    ref = ops.mlp_fused(None, W1, W2, md, **kw)
    guard = 5.0
    hn = torch.full((M + 64, D), guard, device=DEV, dtype=bf16); sn = torch.full((M + 64, 2), guard, device=DEV)
    got = ops.mlp_fused(None, W1, W2, md, next_ln=(ngam, nbet, 1e-6, hn[:M], sn[:M] if mode == "save" else None), **kw)
    for a, b in zip(got, ref):
        assert (a is None and b is None) or torch.equal(a, b)
    assert float((hn[M:].float() - guard).abs().max()) == 0.0 and float((sn[M:] - guard).abs().max()) == 0.0, "rows past M were written"
    y = ref[0]
    h = torch.empty(M, D, device=DEV, dtype=bf16); st = torch.empty(M, 2, device=DEV)
    call("lafs_layernorm_fwd", _p(y), D, _p(ngam), _p(nbet), 1e-6, _p(h), D, None, 0, _p(st), M, D)
    if mode == "save":
        assert torch.equal(sn[:M], st), f"statistics: max {float((sn[:M] - st).abs().max()):.3e}"
    else:
        assert float((sn[:M] - guard).abs().max()) == 0.0, "statistics written without a pointer"
    assert torch.equal(hn[:M], h), f"{int((hn[:M] != h).sum())} of {M * D} operand values differ"
    t = F.layer_norm(y.float().cpu(), (D,), ngam.cpu(), nbet.cpu(), 1e-6)
    assert relerr(hn[:M].float(), t) < 8e-3


@pytest.mark.parametrize("M", [128 * 33, 128 * 33 + 50, 128 * 300 + 64 * 3 + 7])
@pytest.mark.parametrize("mode", ["fwd", "save"])
def test_fused_mlp_with_the_attention_projection_in_front(M, mode):
    """The attention branch's output projection + residual x DropPath (vision_transformer.py:88-90 `self.proj`, :111) as the fused
    MLP's prologue: x1, LayerNorm 2 (operand and statistics), the saved activations and the block's output are BIT-IDENTICAL to
    lafs_gemm_nt(RESID_F32) on the K-resident kernel followed by the fused MLP with the LayerNorm prologue -- whole units, a ragged
    last unit, and more units than CUs (a second launch of 64-row units); rows past M of every output stay untouched."""
    D, H = 384, 1536
    g = torch.Generator().manual_seed(111)
    x0 = (torch.randn(M, D, generator=g) * 1.7 + 0.3).to(DEV)
    o = rnd_bf(M, D, seed=112).to(DEV)
    Wp = rnd_bf(D, D, scale=0.06, seed=113).to(DEV); bp = (torch.randn(D, generator=g) * 0.1).to(DEV)
    gam, bet = (1.0 + 0.2 * torch.randn(D, generator=g)).to(DEV), (0.1 * torch.randn(D, generator=g)).to(DEV)
    W1, W2 = rnd_bf(H, D, scale=0.05, seed=114).to(DEV), rnd_bf(D, H, scale=0.03, seed=115).to(DEV)
    b1, b2 = (torch.randn(H, generator=g) * 0.1).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV)
    nseq = 11
    row2seq = (torch.arange(M) * nseq // M).int().to(DEV)
    sa = torch.tensor([0.0 if i % 4 == 2 else 1.0 / 0.9 for i in range(nseq)]).to(DEV)
    sm = torch.tensor([0.0 if i % 3 == 1 else 1.0 / 0.9 for i in range(nseq)]).to(DEV)
    md = _lib.MLP_FWD if mode == "fwd" else _lib.MLP_FWD_SAVE
    guard = 7.0
    def bufs():
        return (torch.full((M + 64, D), guard, device=DEV), torch.full((M + 64, D), guard, device=DEV, dtype=bf16),
                torch.full((M + 64, 2), guard, device=DEV), torch.full((M + 64, D), guard, device=DEV))
    # ---- separate projection launch
    x1r, h2r, str_, outr = bufs()
    ops.gemm_nt(o, Wp, _lib.EPI_RESID_F32, bias=bp, out=x1r[:M], resid=x0, seq_scale=sa, row2seq=row2seq)
    kw = dict(bias_a=b1, bias_b=b2, seq_scale=sm, row2seq=row2seq, ln=(gam, bet, 1e-6))
    sv = dict(ln_stats=str_[:M], ln_out=h2r[:M]) if mode == "save" else {}
    ref = ops.mlp_fused(None, W1, W2, md, resid=x1r[:M], out=outr[:M], **sv, **kw)
    # ---- in front of the fused MLP
    x1, h2, st, out = bufs()
    sv = dict(ln_stats=st[:M], ln_out=h2[:M]) if mode == "save" else {}
    got = ops.mlp_fused(None, W1, W2, md, resid=x1[:M], out=out[:M], proj=(o, Wp, bp, x0, sa), **sv, **kw)
    assert torch.equal(x1[:M], x1r[:M]), f"x1: {int((x1[:M] != x1r[:M]).sum())} differing values, max {float((x1[:M] - x1r[:M]).abs().max()):.3e}"
    for name, a, b in (("out", out, outr), ("h2", h2, h2r), ("stats", st, str_)):
        assert torch.equal(a, b), f"{name}: {int((a != b).sum())} differing values"          # (rows past M included: the guards)
    assert float((x1[M:] - guard).abs().max()) == 0.0
    for a, b in zip(got[1:], ref[1:]):
        assert (a is None and b is None) or torch.equal(a, b)
    want = x0.cpu() + sa.cpu()[row2seq.cpu().long()][:, None] * (o.float().cpu() @ Wp.float().cpu().t() + bp.cpu())
    assert relerr(x1[:M], want) < 2e-3


@pytest.mark.parametrize("M", [128 * 35, 128 * 33 + 50])
def test_fused_mlp_backward_with_the_layernorm_backward_in_its_epilogue(M):
    """LAFS_MLP_BWD with the LayerNorm-2 backward in the epilogue (vision_transformer.py:112 backward) against the separate path --
    lafs_mlp_fused(BWD) storing dX, then lafs_layernorm_bwd(accumulate) with its partial slots, then the fold: the residual gradient
    stream, its DropPath-scaled bf16 copy and dgamma / dbeta agree to fp32 summation order (the epilogue rounds dX to bf16 exactly
    as the stored tensor is); du is bit-identical; rows past M untouched."""
    D, H = 384, 1536
    g = torch.Generator().manual_seed(91)
    x = (torch.randn(M, D, generator=g) * 1.3 + 0.2).to(DEV)
    gam = (1.0 + 0.2 * torch.randn(D, generator=g)).to(DEV); bet = torch.zeros(D, device=DEV)
    h = torch.empty(M, D, device=DEV, dtype=bf16); st = torch.empty(M, 2, device=DEV)
    call("lafs_layernorm_fwd", _p(x), D, _p(gam), _p(bet), 1e-6, _p(h), D, None, 0, _p(st), M, D)
    W1t, W2t = rnd_bf(D, H, scale=0.05, seed=93).to(DEV), rnd_bf(H, D, scale=0.03, seed=94).to(DEV)     # fc1.weight^T [D, H], fc2.weight^T [H, D]
    dY = rnd_bf(M, D, seed=95).to(DEV)
    gs = (torch.rand(M, H, generator=g) * 1.1).to(bf16).to(DEV)
    nseq = 7
    row2seq = (torch.arange(M) * nseq // M).int().to(DEV)
    sc = torch.tensor([0.0 if i % 3 == 1 else 1.0 / 0.9 for i in range(nseq)]).to(DEV)
    g0 = torch.randn(M, D, generator=g).to(DEV)
    # ---- separate path
    dx, _, du_ref = ops.mlp_fused(dY, W2t, W1t, _lib.MLP_BWD, save_grad=gs)
    g_ref = g0.clone(); gb_ref = torch.empty(M, D, device=DEV, dtype=bf16)
    nparts = int(_lib.lib().lafs_layernorm_bwd_parts(M, D))
    part = torch.zeros(nparts, 2, D, device=DEV)
    call("lafs_layernorm_bwd", _p(dx), D, None, 0, _p(x), D, _p(st), _p(gam), _p(g_ref), D, 1, _p(gb_ref), D, _p(sc), _p(row2seq),
         None, None, M, D, 0.0, 0, None, 0, _p(part))
    dgam_ref, dbet_ref = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
    # ---- one launch
    guard = 9.0
    gio = torch.full((M + 40, D), guard, device=DEV); gio[:M] = g0
    gb = torch.full((M + 40, D), guard, device=DEV, dtype=bf16)
    units = int(_lib.lib().lafs_mlp_fused_ln_parts(M))
    part2 = torch.full((units + 2, 2, D), guard, device=DEV)
    _, _, du = ops.mlp_fused(dY, W2t, W1t, _lib.MLP_BWD, save_grad=gs, ln_bwd=(x, st, gam, gio[:M], gb[:M], part2), seq_scale=sc, row2seq=row2seq)
    assert torch.equal(du, du_ref)
    assert float((gio[M:] - guard).abs().max()) == 0.0 and float((gb[M:].float() - guard).abs().max()) == 0.0 and float((part2[units:] - guard).abs().max()) == 0.0
    e_g = relerr(gio[:M] - g0, g_ref - g0)
    e_gb = relerr(gb[:M].float(), gb_ref.float())
    dgam, dbet = part2[:units, 0].double().sum(0), part2[:units, 1].double().sum(0)
    e_p = max(relerr(dgam, dgam_ref), relerr(dbet, dbet_ref))
    print(f"[mlp-bwd + LN] dx {e_g:.2e}, bf16 copy {e_gb:.2e}, gamma / beta sums {e_p:.2e}")
    assert e_g < 2e-5 and e_gb < 8e-3 and e_p < 2e-5
