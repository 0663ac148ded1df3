"""Tile quantisation of the tiled NT kernel: time per row as the row count crosses whole rounds of 512 workgroup slots
(128x128 tiles, two workgroups per CU).   gpurun -- python tools/lab/t_quant.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lafs_cvpr2024_amd import _lib, ops  # noqa: E402

dev, bf = "cuda", torch.bfloat16


def t_nt(M, N, K, epi, iters=40):
    A = torch.randn(M, K, device=dev).to(bf); B = (torch.randn(N, K, device=dev) * .02).to(bf)
    f32 = epi == _lib.EPI_RESID_F32
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else bf)
    kw = {"resid": torch.randn(M, N, device=dev)} if f32 else {}
    bias = torch.zeros(N, device=dev)
    fn = lambda: ops.gemm_nt(A, B, epi, bias=bias, out=out, **kw)
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


os.environ.setdefault("LAFS_NT_WIDE", "0")
for N, K, epi, name in ((384, 1536, _lib.EPI_RESID_F32, "fc2 fwd (ViT-S)"), (384, 1536, _lib.EPI_BF16, "fc1 dgrad (ViT-S)"),
                        (768, 2048, _lib.EPI_BF16, "fc1 dgrad (Part-fViT)"), (2112, 768, _lib.EPI_BF16, "qkv fwd (Part-fViT)")):
    tn = (N + 127) // 128
    print(f"--- {name}: N={N} K={K}, {tn} column tiles")
    for rounds in (1, 2, 3, 4):
        full = 512 * rounds // tn                       # row tiles that fill `rounds` rounds
        for mt in (full - 1, full, full + 1, full + 4, full + full // (2 * rounds)):
            M = mt * 128
            us = t_nt(M, N, K, epi)
            print(f"  row tiles {mt:4d} (M={M:6d}) tiles {mt*tn:5d} = {mt*tn/512:5.2f} rounds: {us:8.1f} us  {us/mt*1e3:7.1f} ns per row tile  "
                  f"{2*M*N*K/us/1e6:7.1f} TF/s")
