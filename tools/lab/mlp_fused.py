#!/usr/bin/env python3
"""GPU box: the fused MLP kernels (csrc/mlp_fused.hip) against the launches they replace, at the C2 shapes (HIP events, alone).
usage: python tools/lab/mlp_fused.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops

dev = torch.device("cuda", 0)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, M in (("student all rows", 44160), ("student global rows", 25216), ("student local rows", 18944), ("teacher", 25216)):
    D, H = 384, 1536
    x = torch.randn(M, D, device=dev).to(torch.bfloat16)
    w1 = (torch.randn(H, D, device=dev) * 0.05).to(torch.bfloat16); w2 = (torch.randn(D, H, device=dev) * 0.05).to(torch.bfloat16)
    b1, b2 = torch.zeros(H, device=dev), torch.zeros(D, device=dev)
    resid = torch.randn(M, D, device=dev)
    u = torch.empty(M, H, device=dev, dtype=torch.bfloat16); a = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, D, device=dev)
    save = name != "teacher"
    t_f = timeit(lambda: ops.mlp_fwd(x, w1, b1, w2, b2, resid, save=save))
    if save:
        t1 = timeit(lambda: ops.gemm_nt(x, w1, _lib.EPI_BF16_GELU, bias=b1, out=u, out2=a, act=1))
    else:
        t1 = timeit(lambda: ops.gemm_nt(x, w1, _lib.EPI_BF16_GELU, bias=b1, out2=a, skip_pre=True))
    t2 = timeit(lambda: ops.gemm_nt(a, w2, _lib.EPI_RESID_F32, bias=b2, resid=resid, out=out))
    line = f"{name:22s} M={M}: fwd fused {t_f:7.1f} us | fc1 {t1:6.1f} + fc2 {t2:6.1f} = {t1 + t2:6.1f} us"
    if save:
        g = torch.randn(M, D, device=dev).to(torch.bfloat16)
        t_b = timeit(lambda: ops.mlp_bwd(g, w2.t().contiguous(), u, w1.t().contiguous()))
        w2t, w1t = w2.t().contiguous(), w1.t().contiguous()
        t_b = timeit(lambda: ops.mlp_bwd(g, w2t, u, w1t))
        du = torch.empty(M, H, device=dev, dtype=torch.bfloat16); dh = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        t3 = timeit(lambda: ops.gemm_nt(g, w2t, _lib.EPI_DGELU_BF16, aux=u, out=du, act=1))
        t4 = timeit(lambda: ops.gemm_nt(du, w1t, _lib.EPI_BF16, out=dh))
        line += f" || bwd fused {t_b:7.1f} us | dgelu {t3:6.1f} + fc1-dgrad {t4:6.1f} = {t3 + t4:6.1f} us"
    print(line, flush=True)
