#!/usr/bin/env python3
"""Lab: the long-K GEMMs of the ViT-S trunk on the tiled kernel's other tile shapes (lafs_debug_set: 0 library choice = 128x128 tiles
with 64-deep stages; 4 = 256x128 / 64-deep, one workgroup per CU; 6 = 256x128 / 32-deep 3-stage ring, two per CU; 2 = 128x128 / 32-deep; run with LAFS_NT_WIDE=0 to take the single-round 128x384 / 12-wave kernel out of the library choice)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops
sys.argv = ["x", "none"]
exec(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_kernels.py")).read().split("SHAPES = [")[0])
for flag in (0, 4, 6, 2, 0):
    _lib.lib().lafs_debug_set(flag)
    print("debug flags", flag)
    for T_ in (44160, 25216):
        nt(T_, 384, 1536, _lib.EPI_RESID_F32, "fc2 fwd"); nt(T_, 384, 1536, _lib.EPI_BF16, "fc1 dgrad"); nt(T_, 384, 1152, _lib.EPI_BF16, "qkv dgrad")
_lib.lib().lafs_debug_set(0)

# correctness of the lab variant against fp32 torch (RESID and BF16 epilogues, ragged M)
_lib.lib().lafs_debug_set(0)
g = torch.Generator().manual_seed(5)
M, N, K = 128 * 196 + 7, 384, 1536
A = torch.randn(M, K, generator=g).to(torch.bfloat16); B = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
bias = torch.randn(N, generator=g); resid = torch.randn(M, N, generator=g)
ref = A.float() @ B.float().t() + bias
o1 = ops.gemm_nt(A.cuda(), B.cuda(), _lib.EPI_BF16, bias=bias.cuda()).float().cpu()
o2 = ops.gemm_nt(A.cuda(), B.cuda(), _lib.EPI_RESID_F32, bias=bias.cuda(), resid=resid.cuda()).cpu()
print("128x384 / 12-wave kernel (M = 25095): bf16 relerr %.2e  resid relerr %.2e" % (((o1 - ref).abs().max() / ref.abs().max()).item(), ((o2 - (resid + ref)).abs().max() / (resid + ref).abs().max()).item()))
_lib.lib().lafs_debug_set(0)
