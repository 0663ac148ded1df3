"""LayerNorm fused into the residual GEMM's 128x384 epilogue against GEMM + lafs_layernorm_fwd, per launch (HIP events, 200 launches).
    gpurun -- python tools/lab/t_ln_fuse.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops

dev = "cuda"


def timeit(fn, iters=200):
    for _ in range(iters):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for K, name in ((1536, "fc2"), (384, "proj")):
    for M in (25216, 18944, 44160):
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(384, K, device=dev) * 0.02).to(torch.bfloat16)
        b = torch.zeros(384, device=dev); g = torch.ones(384, device=dev); be = torch.zeros(384, device=dev)
        x1 = torch.randn(M, 384, device=dev); out = torch.empty(M, 384, device=dev)
        h = torch.empty(M, 384, device=dev, dtype=torch.bfloat16); st = torch.empty(M, 2, device=dev)
        route = ops.gemm_nt(A, W, _lib.EPI_RESID_F32, bias=b, resid=x1, out=out, route_only=True)
        t_g = timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_RESID_F32, bias=b, resid=x1, out=out))
        t_l = timeit(lambda: ops.layernorm_fwd(out, g, be, 1e-6))
        t_f = timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_RESID_F32, bias=b, resid=x1, out=out, ln=(g, be, 1e-6, h, st)))
        print(f"{name} M={M:6d} K={K:5d}: GEMM (route {route}) {t_g:6.1f} us + ln_fwd {t_l:5.1f} us = {t_g + t_l:6.1f}   fused {t_f:6.1f} us")
