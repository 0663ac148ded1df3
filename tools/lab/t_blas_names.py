"""GPU box, under rocprofv3 --kernel-trace --stats: which hipBLASLt / rocBLAS kernels torch.matmul picks on the step's long-K bf16 GEMM
shapes (the kernel names carry the macro-tile, the MFMA shape and the staging scheme: a design yardstick for lafs_gemm_nt)."""
import torch
dev = "cuda"; torch.manual_seed(0)
for name, M, N, K in (("fc1-dgrad", 44160, 384, 1536), ("fc2-fwd chain1", 25216, 384, 1536), ("Part-fViT qkv", 44160, 2112, 768),
                      ("Part-fViT fc1-dgrad", 44160, 768, 2048), ("Part-fViT fc1 fwd", 44160, 2048, 768), ("square", 8192, 8192, 8192)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(20):
        torch.matmul(A, W.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        torch.matmul(A, W.t(), out=out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e3
    print(f"{name:22s} M={M} N={N} K={K}: {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF/s", flush=True)
