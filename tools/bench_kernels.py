#!/usr/bin/env python3
"""Micro-benchmark of the hot kernels at the C2 shapes (ViT-S, B=64: 44160 student tokens).  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lafs_cvpr2024_amd import _lib, ops

dev = "cuda"
bf = torch.bfloat16


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
    print(f"NT {name:22s} M={M:6d} N={N:5d} K={K:5d}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s  {byts/t/1e9:7.0f} GB/s")


def tn(M, N1, N2, name):
    A = torch.randn(M, N1, device=dev).to(bf); B = torch.randn(M, N2, device=dev).to(bf)
    C = torch.zeros(N1, N2, device=dev)
    t = timeit(lambda: ops.gemm_tn_acc(A, B, C))
    print(f"TN {name:22s} M={M:6d} N1={N1:4d} N2={N2:5d}: {t*1e6:8.1f} us  {2*M*N1*N2/t/1e12:7.1f} TF/s  {(M*(N1+N2)*2)/t/1e9:7.0f} GB/s")


T = 44160
nt(T, 1152, 384, _lib.EPI_BF16, "qkv fwd")
nt(T, 384, 384, _lib.EPI_RESID_F32, "proj fwd")
nt(T, 1536, 384, _lib.EPI_BF16_GELU, "fc1 fwd")
nt(T, 384, 1536, _lib.EPI_RESID_F32, "fc2 fwd")
nt(T, 1536, 384, _lib.EPI_DGELU_BF16, "fc2 dgrad")
nt(T, 384, 1536, _lib.EPI_BF16, "fc1 dgrad")
nt(T, 384, 384, _lib.EPI_BF16, "proj dgrad")
nt(T, 384, 1152, _lib.EPI_BF16, "qkv dgrad")
nt(640, 100096, 256, _lib.EPI_F32, "last layer")
nt(4096, 4096, 4096, _lib.EPI_BF16, "square 4k")
tn(T, 384, 1536, "fc2 wgrad")
tn(T, 1536, 384, "fc1 wgrad")
tn(T, 384, 384, "proj wgrad")
tn(T, 1152, 384, "qkv wgrad")
tn(640, 100096, 256, "last wgrad")

print("--- NT tile height 128 (flag 2) vs 256 (flag 4)")
for flag in (2, 4, 10, 12):
    _lib.lib().lafs_debug_set(flag)
    nt(T, 1152, 384, _lib.EPI_BF16, f"qkv fwd wm{flag}")
    nt(T, 384, 384, _lib.EPI_RESID_F32, f"proj fwd wm{flag}")
    nt(T, 1536, 384, _lib.EPI_BF16_GELU, f"fc1 fwd wm{flag}")
    nt(T, 1536, 384, _lib.EPI_BF16, f"fc1 plain wm{flag}")
    nt(T, 384, 1536, _lib.EPI_RESID_F32, f"fc2 fwd wm{flag}")
    nt(T, 1536, 384, _lib.EPI_DGELU_BF16, f"fc2 dgrad wm{flag}")
    nt(T, 384, 1536, _lib.EPI_BF16, f"fc1 dgrad wm{flag}")
    nt(T, 384, 1152, _lib.EPI_BF16, f"qkv dgrad wm{flag}")
    nt(25216, 1152, 384, _lib.EPI_BF16, f"qkv fwd teacher wm{flag}")
    nt(4096, 4096, 4096, _lib.EPI_BF16, f"square 4k wm{flag}")
_lib.lib().lafs_debug_set(0)

print("--- NT ablations: 0 full, 16 no stores, 32 no mfma, 48 loads only")
for flag in (0, 16, 32, 48):
    _lib.lib().lafs_debug_set(flag)
    nt(T, 1152, 384, _lib.EPI_BF16, f"qkv fwd abl{flag}")
    nt(T, 384, 1536, _lib.EPI_BF16, f"fc1 dgrad abl{flag}")
    nt(T, 1536, 384, _lib.EPI_BF16_GELU, f"fc1 fwd abl{flag}")
_lib.lib().lafs_debug_set(0)

print("--- fc1 GELU epilogue: separate u/a buffers vs one interleaved [T, 2*mlp] buffer")
def gelu_variants():
    M, N, K = T, 1536, 384
    A = torch.randn(M, K, device=dev).to(bf); B = (torch.randn(N, K, device=dev) * .02).to(bf); bias = torch.zeros(N, device=dev)
    u = torch.empty(M, N, device=dev, dtype=bf); a = torch.empty(M, N, device=dev, dtype=bf)
    t = timeit(lambda: ops.gemm_nt(A, B, _lib.EPI_BF16_GELU, bias=bias, out=u, out2=a))
    print(f"   separate buffers      : {t*1e6:7.1f} us")
    ua = torch.empty(M, 2 * N, device=dev, dtype=bf)
    t = timeit(lambda: ops.gemm_nt(A, B, _lib.EPI_BF16_GELU, bias=bias, out=ua[:, :N], out2=ua[:, N:]))
    print(f"   interleaved rows      : {t*1e6:7.1f} us")
    pad = torch.empty(M * N + 4096 + 64, device=dev, dtype=bf)
    a2 = pad[4096 + 64: 4096 + 64 + M * N].view(M, N) if (4096 + 64) % 8 == 0 else a
    t = timeit(lambda: ops.gemm_nt(A, B, _lib.EPI_BF16_GELU, bias=bias, out=u, out2=a2))
    print(f"   separate, a shifted 8K: {t*1e6:7.1f} us")
gelu_variants()
print("--- fc1 GELU ablations: 64 = no second store, 128 = second store without gelu math")
for flag in (0, 64, 128):
    _lib.lib().lafs_debug_set(flag)
    nt(T, 1536, 384, _lib.EPI_BF16_GELU, f"fc1 fwd abl{flag}")
_lib.lib().lafs_debug_set(0)
print("--- loads-only ablation across tile variants (flag 48 + {0: 128x128 bk32, 4: 256x128 bk32, 8: 128x128 bk64, 12: 256x128 bk64})")
for flag in (48, 52, 56, 60):
    _lib.lib().lafs_debug_set(flag)
    nt(T, 1152, 384, _lib.EPI_BF16, f"qkv fwd loads f{flag}")
    nt(T, 384, 1536, _lib.EPI_BF16, f"fc1 dgrad loads f{flag}")
_lib.lib().lafs_debug_set(0)
