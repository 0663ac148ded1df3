"""GPU box: the tiled NT kernel at the eight Part-fViT block shapes (time + output checksums).  The wide-tile lab kernel
(tools/lab/gemm_ntw.hip) was measured with this script while it was wired into lafs_gemm_nt behind LAFS_NTW=0/1 (copy the two files into
csrc/, include gemm_ntw.hpp in gemm.hip and call lafs_ntw_eligible / lafs_ntw_launch in front of the epilogue switch): NOTES.md."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from lafs_cvpr2024_amd import _lib, ops
dev = "cuda"
torch.manual_seed(0)
def timeit(fn, n=100):
    for _ in range(n): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
mode = os.environ.get("LAFS_NTW", "1")
M = int(os.environ.get("LAB_M", "44160"))
lens = [197] * 128 + [37] * 512
row2seq = torch.repeat_interleave(torch.arange(640, device=dev, dtype=torch.int32), torch.tensor(lens, device=dev))[:M].contiguous()
scale = (torch.rand(640, device=dev) > 0.1).float() / 0.9
for name, N, K, epi in (("qkv", 2112, 768, _lib.EPI_BF16), ("fc1", 2048, 768, _lib.EPI_BF16_GELU), ("fc2", 768, 2048, _lib.EPI_RESID_F32),
                        ("proj", 768, 704, _lib.EPI_RESID_F32), ("dgelu", 2048, 768, _lib.EPI_DGELU_BF16), ("dfc1", 768, 2048, _lib.EPI_BF16),
                        ("dqkv", 768, 2112, _lib.EPI_BF16), ("dproj", 704, 768, _lib.EPI_BF16)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    b = torch.randn(N, device=dev) * 0.1
    kw = dict(bias=None if epi == _lib.EPI_DGELU_BF16 else b)
    if epi == _lib.EPI_RESID_F32:
        kw.update(resid=torch.randn(M, N, device=dev), out=torch.empty(M, N, device=dev), seq_scale=scale, row2seq=row2seq, drop_p=0.1, drop_seed=11)
    elif epi == _lib.EPI_BF16_GELU:
        kw.update(out=torch.empty(M, N, device=dev, dtype=torch.bfloat16), out2=torch.empty(M, N, device=dev, dtype=torch.bfloat16), drop_p=0.1, drop_seed=12,
                  act=_lib.ACT_NONE)
    elif epi == _lib.EPI_DGELU_BF16:
        kw.update(aux=torch.randn(M, N, device=dev).to(torch.bfloat16), out=torch.empty(M, N, device=dev, dtype=torch.bfloat16), drop_p=0.1, drop_seed=13)
    else:
        kw.update(out=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
    t = timeit(lambda: ops.gemm_nt(A, W, epi, **kw))
    print(f"LAFS_NTW={mode} {name:6s} M={M} N={N:5d} K={K:5d}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF/s  checksum {float(kw['out'].float().abs().mean()):.6f}"
          + (f" {float(kw['out2'].float().abs().mean()):.6f}" if 'out2' in kw else ""))
