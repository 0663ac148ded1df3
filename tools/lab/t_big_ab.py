"""GPU box: the Part-fViT GEMM shapes on the 128x128 tiled kernel (LAFS_OPT_NT_BIG = 0) against the 192x256 / 256x256 persistent kernel
(gemm_big.hip, forced with LAFS_OPT_NT_BIG = 2 / 3), every epilogue the trunk uses, interleaved on one box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops
dev = "cuda"; torch.manual_seed(0)
ctxs = {k: _lib.Ctx(dev, options={_lib.OPT_NT_BIG: v}, from_env=False) for k, v in (("tiled", 0), ("big1", 2), ("big2", 3), ("slim", 4), ("five", 5), ("auto", 1))}
def timeit(fn, n=60):
    for _ in range(n): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [("fc1 fwd GELU pair", 2048, 768, _lib.EPI_BF16_GELU), ("fc1 fwd GELU only", 2048, 768, "gelu_only"), ("fc1 fwd u + GELU", 2048, 768, "gelu_u"), ("fc2 fwd resid", 768, 2048, _lib.EPI_RESID_F32), ("qkv fwd", 2112, 768, _lib.EPI_BF16),
          ("proj fwd resid", 768, 704, _lib.EPI_RESID_F32), ("dgelu dgrad", 2048, 768, _lib.EPI_DGELU_BF16), ("fc1 dgrad", 768, 2048, _lib.EPI_BF16),
          ("qkv dgrad", 768, 2112, _lib.EPI_BF16), ("proj dgrad", 704, 768, _lib.EPI_BF16)]
for M in (44160, 25216):
    for name, N, K, epi in shapes:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
        tag = epi
        if isinstance(epi, str): epi = _lib.EPI_BF16_GELU
        f32 = epi == _lib.EPI_RESID_F32
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        kw = {}
        if epi == _lib.EPI_BF16_GELU: kw = dict(out2=torch.empty(M, N, device=dev, dtype=torch.bfloat16), act=0 if tag == 'gelu_u' else 1, bias=torch.zeros(N, device=dev), drop_p=0.1, drop_seed=3, skip_pre=(tag == 'gelu_only'))
        if f32: kw = dict(resid=torch.randn(M, N, device=dev), bias=torch.zeros(N, device=dev), drop_p=0.1, drop_seed=3)
        if epi == _lib.EPI_DGELU_BF16: kw = dict(aux=torch.randn(M, N, device=dev).to(torch.bfloat16), act=1, drop_p=0.1, drop_seed=3)
        res = {}
        for rep in range(2):
            for k, c in ctxs.items():
                res.setdefault(k, []).append(timeit(lambda: ops.gemm_nt(A, W, epi, out=out, ctx=c, **kw)))
        fl = 2.0 * M * N * K
        print(f"M={M} {name:18s} N={N} K={K}: " + " | ".join(f"{k} {min(v):6.1f} us ({fl / min(v) / 1e6:5.0f} TF)" for k, v in res.items()), flush=True)
