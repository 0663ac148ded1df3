#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at the C2 shapes (ViT-S, B=64: 44160 student tokens).  GPU box only.
usage: python tools/bench_kernels.py [nt] [tn] [attn] [ablate] [tiles]      (default: nt tn attn)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lafs_cvpr2024_amd import _lib, ops

dev, bf, T = "cuda", torch.bfloat16, 44160
which = set(sys.argv[1:]) or {"nt", "tn", "attn"}


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def nt(M, N, K, epi, name):
    A = torch.randn(M, K, device=dev).to(bf); B = (torch.randn(N, K, device=dev) * .02).to(bf)
    bias = torch.zeros(N, device=dev)
    kw = {}
    f32 = epi in (_lib.EPI_RESID_F32, _lib.EPI_F32)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else bf)
    byts = (M * K + N * K) * 2 + M * N * (4 if f32 else 2)
    if epi == _lib.EPI_BF16_GELU:
        kw["out2"] = torch.empty(M, N, device=dev, dtype=bf); byts += M * N * 2
    if epi == _lib.EPI_RESID_F32:
        kw["resid"] = torch.randn(M, N, device=dev); byts += M * N * 4
    if epi == _lib.EPI_DGELU_BF16:
        kw["aux"] = torch.randn(M, N, device=dev).to(bf); byts += M * N * 2
    t = timeit(lambda: ops.gemm_nt(A, B, epi, bias=None if epi == _lib.EPI_DGELU_BF16 else bias, out=out, **kw))
    print(f"NT {name:24s} M={M:6d} N={N:6d} K={K:5d}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s  {byts/t/1e9:7.0f} GB/s")


def tn(M, N1, N2, name, splits=0):
    A = torch.randn(M, N1, device=dev).to(bf); B = torch.randn(M, N2, device=dev).to(bf)
    C = torch.zeros(N1, N2, device=dev)
    t = timeit(lambda: ops.gemm_tn_acc(A, B, C, splits=splits))
    print(f"TN {name:24s} M={M:6d} N1={N1:6d} N2={N2:5d}: {t*1e6:8.1f} us  {2*M*N1*N2/t/1e12:7.1f} TF/s  {(M*(N1+N2)*2)/t/1e9:7.0f} GB/s")


def wg(M, N1, N2, name):
    A = torch.randn(M, N1, device=dev).to(bf); B = torch.randn(M, N2, device=dev).to(bf)
    C = torch.zeros(N1, N2, device=dev); ws = ops.wgrad_workspace(M, N1, N2, dev); cs = torch.zeros(N1, device=dev)
    t = timeit(lambda: ops.wgrad(A, B, C, accumulate=False, colsum=cs, workspace=ws))
    err = ((C - A.float().t() @ B.float()).abs().max() / C.abs().max()).item()
    print(f"WG {name:24s} M={M:6d} N1={N1:6d} N2={N2:5d}: {t*1e6:8.1f} us  {2*M*N1*N2/t/1e12:7.1f} TF/s  {(M*(N1+N2)*2)/t/1e9:7.0f} GB/s  "
          f"ws {ws.numel()*4/1e6:6.1f} MB  relerr {err:.1e}")


SHAPES = [(T, 1152, 384, _lib.EPI_BF16, "qkv fwd"), (T, 384, 384, _lib.EPI_RESID_F32, "proj fwd"),
          (T, 1536, 384, _lib.EPI_BF16_GELU, "fc1 fwd"), (T, 384, 1536, _lib.EPI_RESID_F32, "fc2 fwd"),
          (T, 1536, 384, _lib.EPI_DGELU_BF16, "fc2 dgrad"), (T, 384, 1536, _lib.EPI_BF16, "fc1 dgrad"),
          (T, 384, 384, _lib.EPI_BF16, "proj dgrad"), (T, 384, 1152, _lib.EPI_BF16, "qkv dgrad"),
          (640, 100096, 256, _lib.EPI_F32, "last layer"), (4096, 4096, 4096, _lib.EPI_BF16, "square 4k")]

if "nt" in which:
    for M, N, K, e, n in SHAPES:
        nt(M, N, K, e, n)
if "tn" in which:
    tn(T, 384, 1536, "fc2 wgrad"); tn(T, 1536, 384, "fc1 wgrad"); tn(T, 384, 384, "proj wgrad"); tn(T, 1152, 384, "qkv wgrad")
    tn(640, 100096, 256, "last wgrad")
    TB = 25216                                                          # ViT-B fine-tune: 128 x 197 tokens
    tn(TB, 768, 2048, "B fc2 wgrad"); tn(TB, 2048, 768, "B fc1 wgrad"); tn(TB, 2112, 768, "B qkv wgrad"); tn(TB, 768, 704, "B proj wgrad")
    for flag, nm in ((1024, "128x128 kb32"), (4096, "256x128"), (2048, "256x256")):
        _lib.lib().lafs_debug_set(flag)
        tn(T, 1536, 384, f"fc1 wgrad {nm}"); tn(TB, 2048, 768, f"B fc1 wgrad {nm}"); tn(TB, 2112, 768, f"B qkv wgrad {nm}")
    _lib.lib().lafs_debug_set(0)
if "wg" in which:
    print("--- wide-tile weight gradient (lafs_wgrad) vs the round-1 kernel (lafs_gemm_tn_acc)")
    TB = 25216
    for (M, N1, N2, name) in ((T, 384, 1536, "fc2 wgrad"), (T, 1536, 384, "fc1 wgrad"), (T, 384, 384, "proj wgrad"), (T, 1152, 384, "qkv wgrad"),
                              (TB, 768, 2048, "B fc2 wgrad"), (TB, 2048, 768, "B fc1 wgrad"), (TB, 2112, 768, "B qkv wgrad"),
                              (TB, 768, 704, "B proj wgrad"), (8192, 4096, 4096, "square")):
        wg(M, N1, N2, name); tn(M, N1, N2, name)
if "wgg" in which:
    print("--- the four weight gradients of one block: 4 x round-1 kernel | 4 x lafs_wgrad | ONE lafs_wgrad_group")
    for (Mx, D, I, H, name) in ((T, 384, 384, 1536, "ViT-S student"), (25216, 384, 384, 1536, "ViT-S teacher-size"), (25216, 768, 704, 2048, "ViT-B fine-tune")):
        mk = lambda r, c: torch.randn(r, c, device=dev).to(bf)
        gbm, a, du, h2, gba, o, dqkv, h1 = mk(Mx, D), mk(Mx, H), mk(Mx, H), mk(Mx, D), mk(Mx, D), mk(Mx, I), mk(Mx, 3 * I), mk(Mx, D)
        pairs = [(gbm, a), (du, h2), (gba, o), (dqkv, h1)]
        Cs = [torch.zeros(x.shape[1], y.shape[1], device=dev) for x, y in pairs]
        bs = [torch.zeros(x.shape[1], device=dev) for x, _ in pairs]
        t0 = timeit(lambda: [ops.gemm_tn_acc(x, y, c, colsum=b) for (x, y), c, b in zip(pairs, Cs, bs)])
        wss = [ops.wgrad_workspace(Mx, x.shape[1], y.shape[1], dev) for x, y in pairs]
        t1 = timeit(lambda: [ops.wgrad(x, y, c, accumulate=True, colsum=b, workspace=w) for (x, y), c, b, w in zip(pairs, Cs, bs, wss)])
        probs = [(x, y, c, True, b) for (x, y), c, b in zip(pairs, Cs, bs)]
        ws = ops.wgrad_group(probs)
        t2 = timeit(lambda: ops.wgrad_group(probs, workspace=ws))
        fl = sum(2 * Mx * x.shape[1] * y.shape[1] for x, y in pairs)
        print(f"   {name:20s} M={Mx}: round-1 {t0*1e6:7.1f} us | 4 x wgrad {t1*1e6:7.1f} us | group {t2*1e6:7.1f} us = {fl/t2/1e12:6.1f} TF/s  (ws {ws.numel()*4/1e6:.1f} MB)")
if "tnsplits" in which:
    print("--- TN wgrad vs number of M-slices (0 = library default)")
    for sp in (0, 8, 16, 24, 32):
        tn(T, 384, 1536, f"fc2 wgrad s{sp}", splits=sp); tn(T, 1536, 384, f"fc1 wgrad s{sp}", splits=sp)
        tn(T, 384, 384, f"proj wgrad s{sp}", splits=sp); tn(T, 1152, 384, f"qkv wgrad s{sp}", splits=sp)
if "tnpart" in which:
    print("--- TN wgrad: device atomics vs per-XCD partial images (+ fold) vs plain stores (wrong results, upper bound)")
    for (M, N1, N2, name) in ((T, 384, 1536, "fc2 wgrad"), (T, 1536, 384, "fc1 wgrad"), (T, 384, 384, "proj wgrad"), (T, 1152, 384, "qkv wgrad")):
        A = torch.randn(M, N1, device=dev).to(bf); B = torch.randn(M, N2, device=dev).to(bf)
        Cd = torch.zeros(N1, N2, device=dev); part = torch.zeros(8, N1, N2, device=dev)
        ops.gemm_tn_acc(A, B, Cd)
        ops.gemm_tn_part(A, B, part); Cp = ops.reduce_partials(part, torch.zeros(N1, N2, device=dev))
        ref = A.float().t() @ B.float()
        print(f"   {name}: max|atomics-ref| {float((Cd-ref).abs().max()):.3e}  max|partials-ref| {float((Cp-ref).abs().max()):.3e}  (|ref| {float(ref.abs().max()):.1f})"
              f"  images left zero: {float(part.abs().max()) == 0.0}")
        t0 = timeit(lambda: ops.gemm_tn_acc(A, B, Cd))
        t1 = timeit(lambda: ops.gemm_tn_part(A, B, part))
        out = torch.zeros(N1, N2, device=dev)
        t2 = timeit(lambda: ops.reduce_partials(part, out))
        _lib.lib().lafs_debug_set(1); t3 = timeit(lambda: ops.gemm_tn_acc(A, B, Cd)); _lib.lib().lafs_debug_set(0)
        print(f"   {name}: atomics {t0*1e6:7.1f} us | partial {t1*1e6:7.1f} us + fold {t2*1e6:6.1f} us | stores {t3*1e6:7.1f} us")
if "stagger" in which:
    print("--- NT GEMMs (LAFS_USE_ABLATE_LIB=1): second round-robin slot of each CU held back by n x ~4 us; 16 = no stores, 32 = no MFMA")
    for flag in (0, 1 << 20, 2 << 20, 3 << 20, 16, 32, 48):
        _lib.lib().lafs_debug_set(flag)
        for M, N, K, e, n in SHAPES[:8]:
            nt(M, N, K, e, f"{n} f{flag}")
    _lib.lib().lafs_debug_set(0)
if "gelucost" in which:
    print("--- fc1 forward (LAFS_USE_ABLATE_LIB=1): 0 product | 128 second tensor stored without the GELU math | 64 no second store")
    for flag in (16, 16 + 2, 16 + 2 + 8, 16 + 4 + 8, 48, 48 + 2, 48 + 2 + 8, 48 + 4 + 8):
        _lib.lib().lafs_debug_set(flag)
        nt(T, 1536, 384, _lib.EPI_BF16_GELU, f"fc1 fwd f{flag}")
        nt(25216, 1536, 384, _lib.EPI_BF16_GELU, f"fc1 fwd teacher-size f{flag}")
    _lib.lib().lafs_debug_set(0)
if "nttile" in which:
    print("--- NT GEMMs (LAFS_USE_ABLATE_LIB=1): 0 library choice | 2 128x128 tiles | 4 256x128 tiles | 8 64-deep stages")
    for flag in (0, 65536, 65536 + 8):
        _lib.lib().lafs_debug_set(flag)
        for M, N, K, e, n in SHAPES[:8]:
            nt(M, N, K, e, f"{n} f{flag}")
    _lib.lib().lafs_debug_set(0)
if "ntstore" in which:
    print("--- NT epilogue stores: normal (default) vs non-temporal (flag 256)")
    for flag in (0, 256):
        _lib.lib().lafs_debug_set(flag)
        for M, N, K, e, n in SHAPES[:8]:
            nt(M, N, K, e, f"{n} f{flag}")
    _lib.lib().lafs_debug_set(0)
if "mlp" in which:
    print("--- fc1 (+GELU) -> fc2 (+residual) back to back, student shape; flags: 0 normal stores, 256 non-temporal stores")
    A = torch.randn(T, 384, device=dev).to(bf); W1 = (torch.randn(1536, 384, device=dev) * .02).to(bf); W2 = (torch.randn(384, 1536, device=dev) * .02).to(bf)
    b1 = torch.zeros(1536, device=dev); b2 = torch.zeros(384, device=dev)
    # distinct buffers per "layer" so that the working set does not sit in the Infinity Cache between iterations
    L = 6
    us = [torch.empty(T, 1536, device=dev, dtype=bf) for _ in range(L)]; as_ = [torch.empty(T, 1536, device=dev, dtype=bf) for _ in range(L)]
    xs = [torch.randn(T, 384, device=dev) for _ in range(L + 1)]
    def chain():
        for l in range(L):
            ops.gemm_nt(A, W1, _lib.EPI_BF16_GELU, bias=b1, out=us[l], out2=as_[l])
            ops.gemm_nt(as_[l], W2, _lib.EPI_RESID_F32, bias=b2, resid=xs[l], out=xs[l + 1])
    for flag in (0, 256):
        _lib.lib().lafs_debug_set(flag)
        t = timeit(chain, iters=10)
        print(f"   flag {flag:4d}: {t / L * 1e6:8.1f} us per fc1+fc2 pair")
    _lib.lib().lafs_debug_set(0)
TB = 25216
SHAPES_B = [(TB, 2112, 768, _lib.EPI_BF16, "B qkv fwd"), (TB, 768, 704, _lib.EPI_RESID_F32, "B proj fwd"),
            (TB, 2048, 768, _lib.EPI_BF16_GELU, "B fc1 fwd"), (TB, 768, 2048, _lib.EPI_RESID_F32, "B fc2 fwd"),
            (TB, 2048, 768, _lib.EPI_DGELU_BF16, "B fc2 dgrad"), (TB, 768, 2048, _lib.EPI_BF16, "B fc1 dgrad"),
            (TB, 704, 768, _lib.EPI_BF16, "B proj dgrad"), (TB, 768, 2112, _lib.EPI_BF16, "B qkv dgrad")]
if "tilesb" in which:
    print("--- ViT-B (fine-tune) NT tile variants: 0 = heuristic, 2 = 128x128 bk32, 4 = 256x128 bk32, 10 = 128x128 bk64, 12 = 256x128 bk64")
    for flag in (0, 2, 4, 10, 12):
        _lib.lib().lafs_debug_set(flag)
        for M, N, K, e, n in SHAPES_B:
            if (flag & 8) and K % 64:
                continue
            nt(M, N, K, e, f"{n} f{flag}")
    _lib.lib().lafs_debug_set(0)
if "interleave" in which:
    print("--- fc1: separate u / GELU(u) buffers vs 64-byte interleaved in one [M, 2N] buffer (flag 16384)")
    A = torch.randn(T, 384, device=dev).to(bf); W1 = (torch.randn(1536, 384, device=dev) * .02).to(bf); b1 = torch.zeros(1536, device=dev)
    L = 6
    bufs = [torch.empty(T, 3072, device=dev, dtype=bf) for _ in range(L)]
    def chain():
        for l in range(L):
            ops.gemm_nt(A, W1, _lib.EPI_BF16_GELU, bias=b1, out=bufs[l][:, :1536], out2=bufs[l][:, 1536:])
    def chain_i():
        for l in range(L):
            ops.gemm_nt(A, W1, _lib.EPI_BF16_GELU, bias=b1, out=bufs[l].view(-1)[:T * 1536].view(T, 1536), out2=bufs[l].view(-1)[T * 1536:].view(T, 1536))
    _lib.lib().lafs_debug_set(32768)
    tl = timeit(chain_i, iters=10) / L
    _lib.lib().lafs_debug_set(0)
    print(f"   piece-by-piece order (flag 32768), two buffers: {tl*1e6:7.1f} us")
    t0 = timeit(chain, iters=10) / L
    t0b = timeit(chain_i, iters=10) / L
    _lib.lib().lafs_debug_set(16384)
    t1 = timeit(chain_i, iters=10) / L
    _lib.lib().lafs_debug_set(0)
    print(f"   halves of one [M,3072] row: {t0*1e6:7.1f} us | two [M,1536] buffers: {t0b*1e6:7.1f} us | 64-B interleaved: {t1*1e6:7.1f} us")
if "nt256" in which:
    print("--- NT 256x256 tiles (flag 65536) vs the heuristic choice, wide-output shapes")
    wide = [x for x in SHAPES[:8] + SHAPES_B if x[1] >= 1024 and x[3] in (_lib.EPI_BF16, _lib.EPI_BF16_GELU, _lib.EPI_DGELU_BF16)]
    for flag in (0, 65536):
        _lib.lib().lafs_debug_set(flag)
        for M, N, K, e, n in wide:
            nt(M, N, K, e, f"{n} f{flag}")
    _lib.lib().lafs_debug_set(0)
if "augment" in which:
    from lafs_cvpr2024_amd.augment import DeviceAugmenter
    da = DeviceAugmenter(64, n_local=8, device=dev, seed=0)
    u8 = torch.randint(0, 256, (64, 3, 112, 112), device=dev, dtype=torch.uint8)
    t = timeit(lambda: da(u8), iters=20)                     # vectorised host sampling + parameter upload + one launch
    from lafs_cvpr2024_amd.ops import _p, call
    tk = timeit(lambda: call("lafs_augment_views", _p(u8), _p(da.params_dev), _p(da.table), 64, 10, _p(da.views)), iters=20)
    print(f"--- device augmentation: 64 images -> 1280 views: {t*1e6:8.1f} us per batch end to end, kernel alone {tk*1e6:8.1f} us "
          f"({1280/tk/1e6:.2f} M views/s)")
if "tiles" in which:
    print("--- NT tile variants: flag 2 = 128x128 bk32, 4 = 256x128 bk32, 10 = 128x128 bk64, 12 = 256x128 bk64")
    for flag in (2, 4, 10, 12):
        _lib.lib().lafs_debug_set(flag)
        for M, N, K, e, n in SHAPES[:8]:
            nt(M, N, K, e, f"{n} f{flag}")
    _lib.lib().lafs_debug_set(0)
if "ablate" in which:
    print("--- NT ablations: 0 full, 16 no stores, 32 no mfma, 48 loads only, 64 no second (GELU) store")
    for flag in (0, 16, 32, 48, 64, 128, 192):
        _lib.lib().lafs_debug_set(flag)
        nt(T, 1152, 384, _lib.EPI_BF16, f"qkv fwd abl{flag}")
        nt(T, 1536, 384, _lib.EPI_BF16_GELU, f"fc1 fwd abl{flag}")
    for flag in (2, 4):
        _lib.lib().lafs_debug_set(flag)
        nt(T, 1536, 384, _lib.EPI_BF16_GELU, f"fc1 fwd tile f{flag}")
        nt(T, 1536, 384, _lib.EPI_DGELU_BF16, f"fc2 dgrad tile f{flag}")
    _lib.lib().lafs_debug_set(1)
    print("--- TN with plain stores instead of atomics (flag 1)")
    tn(T, 384, 1536, "fc2 wgrad")
    _lib.lib().lafs_debug_set(0)
if "ln" in which:
    print("--- LayerNorm forward / backward at the student shape (44160 x 384); distinct buffers per call (no cache residency)")
    L = 6
    xs = [torch.randn(T, 384, device=dev) for _ in range(L)]; dys = [torch.randn(T, 384, device=dev).to(bf) for _ in range(L)]
    gs = [torch.randn(T, 384, device=dev) for _ in range(L)]; gbs = [torch.empty(T, 384, device=dev, dtype=bf) for _ in range(L)]
    gam = torch.ones(384, device=dev); bet = torch.zeros(384, device=dev); dgam = torch.zeros(384, device=dev); dbet = torch.zeros(384, device=dev)
    outs = [ops.layernorm_fwd(x, gam, bet, 1e-6) for x in xs]
    tf = timeit(lambda: [ops.layernorm_fwd(x, gam, bet, 1e-6) for x in xs]) / L
    tb = timeit(lambda: [ops.layernorm_bwd(dys[i], xs[i], outs[i][2], gam, gs[i], dgam, dbet, accumulate=True, gb_out=gbs[i]) for i in range(L)]) / L
    print(f"   fwd {tf*1e6:6.1f} us ({T*384*6/tf/1e9:6.0f} GB/s)   bwd {tb*1e6:6.1f} us ({T*384*16/tb/1e9:6.0f} GB/s)")
if "dzn" in which:
    print("--- head backward dzn = dlogits Wn (M=640, N=256, K=100096): split-K atomics vs slice images + fold")
    from lafs_cvpr2024_amd.ops import _p, call
    A = torch.randn(640, 100096, device=dev).to(bf); B = torch.randn(256, 100096, device=dev).to(bf)
    for sp in (16, 32, 64, 96):
        ta = timeit(lambda: ops.gemm_nt(A, B, _lib.EPI_ATOMIC_F32, splits=sp))
        ns = _lib.lib().lafs_gemm_nt_slices(100096, sp)
        part = torch.empty(ns, 640, 256, device=dev); out = torch.empty(640, 256, device=dev)
        def f():
            ops.gemm_nt(A, B, _lib.EPI_F32, splits=sp, out=part.view(-1, 256), out_rows=ns * 640)
            call("lafs_sum_slices", _p(part), 640 * 256, ns, 640 * 256, _p(out))
        tb = timeit(f)
        ref = ops.gemm_nt(A, B, _lib.EPI_ATOMIC_F32, splits=sp)
        print(f"   splits {sp:3d}: atomics {ta*1e6:6.1f} us | images + fold {tb*1e6:6.1f} us   max diff {float((ref - out).abs().max()):.2e} (|ref| {float(ref.abs().max()):.1f})")
if "attn" in which:
    print("--- attention (student shapes: 128 seqs x 197 and 512 x 37, 6 heads)")
    for nseq, n in ((128, 197), (512, 37)):
        heads = 6
        cu = torch.arange(0, (nseq + 1) * n, n, dtype=torch.int32, device=dev)
        qkv = torch.randn(nseq * n, 3 * heads * 64, device=dev).to(bf)
        out, lse = ops.attention_fwd(qkv, cu, n, heads, 0.125)
        dout = torch.randn(nseq * n, heads * 64, device=dev).to(bf)
        tf = timeit(lambda: ops.attention_fwd(qkv, cu, n, heads, 0.125))
        tb = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, cu, n, heads, 0.125))
        fl = 4 * nseq * heads * n * n * 64
        print(f"   {nseq:4d} x {n:3d}: fwd {tf*1e6:7.1f} us ({fl/tf/1e12:6.1f} TF/s)   bwd {tb*1e6:7.1f} us ({2.5*fl/tb/1e12:6.1f} TF/s)")
