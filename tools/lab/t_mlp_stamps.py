"""Lab: phase stamps of the fused MLP kernel (tools/lab/libmlp_abl128.so: wave 0 of every workgroup, s_memtime at 100 MHz).
usage: python tools/lab/t_mlp_stamps.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib

DEV, bf16 = "cuda", torch.bfloat16
D, H = 384, 1536
here = os.path.dirname(os.path.abspath(__file__))
h = C.CDLL(os.path.join(here, "libmlp_abl128.so"))
fn = h.lafs_mlp_fused
fn.argtypes = [C.POINTER(_lib.MlpArgs), C.c_void_p]; fn.restype = C.c_int
names = ["wait+barrier A", "DMA issue A", "MFMA A", "mid-epilogue", "wait+barrier B", "DMA issue B (+g fetch)", "MFMA B", "prologue", "final epilogue", "whole kernel"]
for M in (25216, 128 * 256):
    g = torch.Generator().manual_seed(1)
    X = torch.randn(M, D, generator=g).to(bf16).to(DEV)
    W1 = (torch.randn(H, D, generator=g) * 0.05).to(bf16).to(DEV); W2 = (torch.randn(D, H, generator=g) * 0.03).to(bf16).to(DEV)
    b1, b2 = torch.randn(H, generator=g).to(DEV) * 0.1, torch.randn(D, generator=g).to(DEV) * 0.1
    resid = torch.randn(M, D, generator=g).to(DEV); out = torch.empty(M, D, device=DEV); dx = torch.empty(M, D, device=DEV, dtype=bf16)
    gs = torch.rand(M, H, device=DEV).to(bf16); a_ = torch.empty(M, H, device=DEV, dtype=bf16)
    gam, bet = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    units = (M + 127) // 128
    for mode, ln in ((0, False), (0, True), (1, True), (2, False)):
        stamps = torch.zeros(units * 10, dtype=torch.int64, device=DEV)
        a = _lib.MlpArgs()
        a.X, a.ldx, a.Wa, a.ldwa, a.Wb, a.ldwb = X.data_ptr(), D, W1.data_ptr(), D, W2.data_ptr(), H
        a.M, a.H, a.mode = M, H, mode
        a.bias_a, a.bias_b, a.resid, a.ldr = b1.data_ptr(), b2.data_ptr(), resid.data_ptr(), D
        a.out, a.ldo = (out.data_ptr() if mode != 2 else dx.data_ptr()), D
        a.save_grad, a.ldsg, a.save_act, a.ldsa = gs.data_ptr(), H, a_.data_ptr(), H
        if ln:
            a.ln_gamma, a.ln_beta, a.ln_eps = gam.data_ptr(), bet.data_ptr(), 1e-6
        a.ctx = stamps.data_ptr()
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(20):
            fn(C.byref(a), C.c_void_p(st))
        torch.cuda.synchronize()
        t = stamps.view(units, 10).double().cpu() * 10.0 / 1e3          # us (100 MHz ticks)
        med = t.median(dim=0).values
        print(f"M={M} mode {mode} ln {int(ln)}: " + "  ".join(f"{n} {v:.1f}" for n, v in zip(names, med.tolist())) +
              f"   | slowest workgroup {t[:, 9].max():.1f} us, fastest {t[:, 9].min():.1f}", flush=True)
