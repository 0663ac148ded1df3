"""GPU box: fc2 forward (M = 44160, N = 384, K = 1536, RESID_F32) with and without the DropPath scale lookup in its epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops
dev = "cuda"
M, N, K = 44160, 384, 1536
A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
b = torch.zeros(N, device=dev); x1 = torch.randn(M, N, device=dev); out = torch.empty(M, N, device=dev)
lens = [197] * 128 + [37] * 512
row2seq = torch.repeat_interleave(torch.arange(640, device=dev, dtype=torch.int32), torch.tensor(lens, device=dev))
scale = (torch.rand(640, device=dev) > 0.1).float() / 0.9
def timeit(fn, n=300):
    for _ in range(n): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(2):
    print("plain      %.1f us" % timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_RESID_F32, bias=b, resid=x1, out=out)))
    print("seq_scale  %.1f us" % timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_RESID_F32, bias=b, resid=x1, out=out, seq_scale=scale, row2seq=row2seq)))
    print("in place   %.1f us" % timeit(lambda: ops.gemm_nt(A, W, _lib.EPI_RESID_F32, bias=b, resid=x1, out=x1, seq_scale=scale, row2seq=row2seq)))
