"""GPU box, under tools/lab/pmc.sh / pmc_mem.sh: the Part-fViT GEMM shapes at the `mynet` pair's 44 160 student rows on the default routes
(gemm_big.hip for the plain / GELU-pair / residual / GELU' epilogues), 20 launches each, for the per-kernel counter tables."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops
dev = "cuda"; torch.manual_seed(0); M = 44160
shapes = [("fc1 fwd GELU pair", 2048, 768, _lib.EPI_BF16_GELU), ("fc2 fwd resid", 768, 2048, _lib.EPI_RESID_F32), ("qkv fwd", 2112, 768, _lib.EPI_BF16),
          ("dgelu dgrad", 2048, 768, _lib.EPI_DGELU_BF16), ("fc1 dgrad", 768, 2048, _lib.EPI_BF16), ("qkv dgrad", 768, 2112, _lib.EPI_BF16)]
for name, N, K, epi in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    f32 = epi == _lib.EPI_RESID_F32
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    kw = {}
    if epi == _lib.EPI_BF16_GELU: kw = dict(out2=torch.empty(M, N, device=dev, dtype=torch.bfloat16), act=1, bias=torch.zeros(N, device=dev), drop_p=0.1, drop_seed=3)
    if f32: kw = dict(resid=torch.randn(M, N, device=dev), bias=torch.zeros(N, device=dev), drop_p=0.1, drop_seed=3)
    if epi == _lib.EPI_DGELU_BF16: kw = dict(aux=torch.randn(M, N, device=dev).to(torch.bfloat16), act=1, drop_p=0.1, drop_seed=3)
    for _ in range(20): ops.gemm_nt(A, W, epi, out=out, **kw)
    torch.cuda.synchronize()
    print(name, "route", ops.gemm_nt(A, W, epi, out=out, route_only=True, **kw), flush=True)
