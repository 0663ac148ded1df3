import sys, os
sys.path.insert(0, os.getcwd())
import torch
from lafs_cvpr2024_amd import ops
sys.argv=['x']
dev='cuda'; bf=torch.bfloat16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
A=torch.randn(640,100096,device=dev).to(bf); B=torch.randn(640,256,device=dev).to(bf)
C=torch.zeros(100096,256,device=dev)
print("old tn_acc splits=1 (+zero fill):", timeit(lambda: (C.zero_(), ops.gemm_tn_acc(A,B,C,splits=1))), "us")
ws=ops.wgrad_workspace(640,100096,256,dev)
print("wgrad direct overwrite:", timeit(lambda: ops.wgrad(A,B,C,accumulate=False,workspace=ws)), "us")
