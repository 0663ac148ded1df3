"""GPU box: torch.matmul (rocBLAS / hipBLASLt) on the plain bf16 GEMM shapes of the ViT-S step, beside lafs_gemm_nt -- a yardstick."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops
dev = "cuda"; torch.manual_seed(0)
def timeit(fn, n=200):
    for _ in range(n): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, M, N, K in (("fc1-dgrad", 44160, 384, 1536), ("qkv-dgrad", 44160, 384, 1152), ("proj-dgrad", 44160, 384, 384), ("qkv fwd", 44160, 1152, 384),
                      ("fc1-dgrad chain1", 25216, 384, 1536), ("fc1-dgrad chain2", 18944, 384, 1536), ("fc2-dgrad (no GELU')", 44160, 1536, 384),
                      ("Part-fViT qkv", 44160, 2112, 768), ("Part-fViT fc1-dgrad", 44160, 768, 2048)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    Wt = W.t().contiguous()
    t_lafs = timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_BF16, out=out))
    t_nt = timeit(lambda: torch.matmul(A, W.t(), out=out))
    t_nn = timeit(lambda: torch.matmul(A, Wt, out=out))
    fl = 2.0 * M * N * K
    print(f"{name:22s} M={M} N={N} K={K}: lafs_gemm_nt {t_lafs:6.1f} us ({fl / t_lafs / 1e6:5.0f} TF/s) | torch NT {t_nt:6.1f} us ({fl / t_nt / 1e6:5.0f}) | torch NN {t_nn:6.1f} us ({fl / t_nn / 1e6:5.0f})")
