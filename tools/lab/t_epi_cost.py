"""GPU box: what the epilogue arithmetic of the tiled NT kernel costs at the Part-fViT fc1 / GELU'-dgrad / fc2 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops
dev = "cuda"; torch.manual_seed(0)
def timeit(fn, n=100):
    for _ in range(n): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = int(os.environ.get("LAB_M", "44160"))
N, K = 2048, 768
A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16); b = torch.randn(N, device=dev) * 0.1
o1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16); o2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16); aux = torch.randn(M, N, device=dev).to(torch.bfloat16)
SG = 1   # LAFS_GELU_SAVE_GRAD
print("plain bf16 output          %.1f us" % timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_BF16, bias=b, out=o1)))
print("GELU, u + gelu(u)          %.1f us" % timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_BF16_GELU, bias=b, out=o1, out2=o2)))
print("GELU, gelu(u) only         %.1f us" % timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_BF16_GELU, bias=b, out=o1, out2=o2, skip_pre=True)))
print("GELU + dropout             %.1f us" % timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_BF16_GELU, bias=b, out=o1, out2=o2, drop_p=0.1, drop_seed=3)))
print("GELU' saved + gelu         %.1f us" % timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_BF16_GELU, bias=b, out=o1, out2=o2, act=SG)))
print("GELU' saved + gelu + drop  %.1f us" % timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_BF16_GELU, bias=b, out=o1, out2=o2, act=SG, drop_p=0.1, drop_seed=3)))
print("dGELU (aux = u)            %.1f us" % timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_DGELU_BF16, aux=aux, out=o1)))
print("dGELU (aux = gelu')        %.1f us" % timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_DGELU_BF16, aux=aux, out=o1, act=SG)))
print("dGELU (aux = gelu') + drop %.1f us" % timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_DGELU_BF16, aux=aux, out=o1, act=SG, drop_p=0.1, drop_seed=3)))
