"""GPU box: the fused head + DINO loss (csrc/dino_head_loss.hip) against the unfused launches it replaces, C2 sizes, per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops
from lafs_cvpr2024_amd.ops import _p, call
dev = "cuda"; torch.manual_seed(0)
ncrops, B, K = 10, 64, 100000
Kpad = (K + 127) // 128 * 128
nrm = lambda t: torch.nn.functional.normalize(t, dim=1)
zs, zt = nrm(torch.randn(ncrops * B, 256, device=dev)).bfloat16(), nrm(torch.randn(2 * B, 256, device=dev)).bfloat16()
ws, wt = nrm(torch.randn(Kpad, 256, device=dev)).bfloat16(), nrm(torch.randn(Kpad, 256, device=dev)).bfloat16()
center = 0.05 * torch.randn(K, device=dev); temps = torch.tensor([0.1, 0.04], device=dev)
ls = torch.empty(ncrops * B, Kpad, device=dev); lt = torch.empty(2 * B, Kpad, device=dev)
grad = torch.zeros(ncrops * B, Kpad, device=dev, dtype=torch.bfloat16); col = torch.empty(K, device=dev); loss = torch.empty(1, device=dev)
ws1 = torch.empty(_lib.lib().lafs_dino_loss_workspace(ncrops, B, K), device=dev); ws2 = torch.empty(_lib.lib().lafs_dino_head_loss_workspace(ncrops, B, K), device=dev)
def unfused():
    ops.gemm_nt(zs, ws, _lib.EPI_F32, out=ls, n_cols=Kpad); ops.gemm_nt(zt, wt, _lib.EPI_F32, out=lt, n_cols=Kpad)
    ops.dino_loss_fwd_bwd(ls, lt, center, ncrops, 0.1, 0.04, K=K, grad=grad, ws=ws1, loss=loss, dev_temps=temps)
    call("lafs_colsum_f32", _p(lt), Kpad, 2 * B, K, _p(col))
def fused():
    ops.dino_head_loss(zs, zt, ws, wt, center, ncrops, K, 0.1, 0.04, grad=grad, loss=loss, colsum=col, ws=ws2, dev_temps=temps)
def timeit(fn, n=100):
    for _ in range(n): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(2):
    print(f"unfused {timeit(unfused):7.1f} us | fused {timeit(fused):7.1f} us", flush=True)
