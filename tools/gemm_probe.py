#!/usr/bin/env python3
"""Run ONE GEMM shape a few times (for rocprofv3 --pmc).  usage: gemm_probe.py M N K epi iters"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lafs_cvpr2024_amd import _lib, ops
M, N, K, epi, iters = (int(x) for x in sys.argv[1:6])
dev = "cuda"; bf = torch.bfloat16
A = torch.randn(M, K, device=dev).to(bf); B = (torch.randn(N, K, device=dev) * .02).to(bf)
f32 = epi in (_lib.EPI_RESID_F32, _lib.EPI_F32)
out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else bf)
kw = {}
if epi == _lib.EPI_BF16_GELU: kw["out2"] = torch.empty(M, N, device=dev, dtype=bf)
if epi == _lib.EPI_RESID_F32: kw["resid"] = torch.randn(M, N, device=dev)
if epi == _lib.EPI_DGELU_BF16: kw["aux"] = torch.randn(M, N, device=dev).to(bf)
for _ in range(iters):
    ops.gemm_nt(A, B, epi, bias=None if epi == 4 else torch.zeros(N, device=dev), out=out, **kw)
torch.cuda.synchronize()
