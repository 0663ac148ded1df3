"""Lab: lafs_mlp_fused against the two lafs_gemm_nt launches it replaces, at the C2 row counts (HIP events, interleaved).
usage: python tools/lab/t_mlp_fused.py [reps]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops

DEV, bf16 = "cuda", torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
D, H = 384, 1536


def timeit(fn, n=reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M in (25216, 44160, 18944):
    g = torch.Generator().manual_seed(1)
    X = (torch.randn(M, D, generator=g)).to(bf16).to(DEV)
    W1 = (torch.randn(H, D, generator=g) * 0.05).to(bf16).to(DEV); W2 = (torch.randn(D, H, generator=g) * 0.03).to(bf16).to(DEV)
    W2t, W1t = W2.t().contiguous(), W1.t().contiguous()
    b1, b2 = torch.randn(H, generator=g).to(DEV) * 0.1, torch.randn(D, generator=g).to(DEV) * 0.1
    resid = torch.randn(M, D, generator=g).to(DEV)
    out = torch.empty(M, D, device=DEV)
    a = torch.empty(M, H, device=DEV, dtype=bf16); gs = torch.empty(M, H, device=DEV, dtype=bf16); du = torch.empty(M, H, device=DEV, dtype=bf16)
    dx = torch.empty(M, D, device=DEV, dtype=bf16)
    dY = torch.randn(M, D, generator=g).to(bf16).to(DEV)
    row2seq = (torch.arange(M) * 64 // M).int().to(DEV); sc = torch.full((64,), 1.0 / 0.9, device=DEV)
    kw = dict(resid=resid, seq_scale=sc, row2seq=row2seq)

    def two_t():
        ops.gemm_nt(X, W1, _lib.EPI_BF16_GELU, bias=b1, out2=a, skip_pre=True)
        ops.gemm_nt(a, W2, _lib.EPI_RESID_F32, bias=b2, out=out, **kw)

    def two_s():
        ops.gemm_nt(X, W1, _lib.EPI_BF16_GELU, bias=b1, out=gs, out2=a, act=1)
        ops.gemm_nt(a, W2, _lib.EPI_RESID_F32, bias=b2, out=out, **kw)

    def two_b():
        ops.gemm_nt(dY, W2t, _lib.EPI_DGELU_BF16, aux=gs, act=1, out=du)
        ops.gemm_nt(du, W1t, _lib.EPI_BF16, out=dx)

    fus_t = lambda: ops.mlp_fused(X, W1, W2, _lib.MLP_FWD, bias_a=b1, bias_b=b2, out=out, **kw)
    fus_s = lambda: ops.mlp_fused(X, W1, W2, _lib.MLP_FWD_SAVE, bias_a=b1, bias_b=b2, out=out, save_grad=gs, save_act=a, **kw)
    fus_b = lambda: ops.mlp_fused(dY, W2t, W1t, _lib.MLP_BWD, out=dx, save_grad=gs, save_act=du)
    two_s()
    res = {}
    for rnd in range(2):
        for name, fn in (("two_t", two_t), ("fus_t", fus_t), ("two_s", two_s), ("fus_s", fus_s), ("two_b", two_b), ("fus_b", fus_b)):
            res.setdefault(name, []).append(timeit(fn))
    fl = 2 * 2 * M * D * H
    print(f"M={M}: " + "  ".join(f"{k} {min(v):6.1f} us ({fl / min(v) * 1e-6:5.0f} TF/s)" for k, v in res.items()), flush=True)

# ---- ablation variants of the fused kernel (tools/lab/libmlp_abl<N>.so, `make -C tools/lab mlp_abl`): teacher forward at 25 216 rows
import ctypes as C
import glob
M = 25216
g = torch.Generator().manual_seed(1)
X = (torch.randn(M, D, generator=g)).to(bf16).to(DEV)
W1 = (torch.randn(H, D, generator=g) * 0.05).to(bf16).to(DEV); W2 = (torch.randn(D, H, generator=g) * 0.03).to(bf16).to(DEV)
b1, b2 = torch.randn(H, generator=g).to(DEV) * 0.1, torch.randn(D, generator=g).to(DEV) * 0.1
resid = torch.randn(M, D, generator=g).to(DEV); out = torch.empty(M, D, device=DEV)
gs = torch.empty(M, H, device=DEV, dtype=bf16); a_ = torch.empty(M, H, device=DEV, dtype=bf16)
names = {32: "half the stage-B reads", 64: "half the stage-A reads", 96: "half of all reads", 97: "half the reads, no GELU", 0: "product", 1: "no GELU math", 2: "no fragment reads", 4: "no DMA", 8: "no MFMA", 16: "no barriers", 3: "no GELU, no reads", 7: "MFMA + barriers only"}
here = os.path.dirname(os.path.abspath(__file__))
libs = sorted(glob.glob(os.path.join(here, "libmlp_abl*.so")), key=lambda s: int(s.split("abl")[-1][:-3])) + sorted(glob.glob(os.path.join(here, "libmlp_fd*.so")))
runs = []
for path in [None] + libs:
    n = 0 if path is None else (int(path.split("abl")[-1][:-3]) if "abl" in path else 1000 + int(path.split("_fd")[-1][:-3]))
    if n >= 1000:
        names[n] = "FD = %d" % (n - 1000)
    h = _lib.lib() if path is None else C.CDLL(path)
    fn = h.lafs_mlp_fused
    fn.argtypes = [C.POINTER(_lib.MlpArgs), C.c_void_p]; fn.restype = C.c_int
    for mode in (_lib.MLP_FWD, _lib.MLP_FWD_SAVE):
        a = _lib.MlpArgs()
        a.X, a.ldx, a.Wa, a.ldwa, a.Wb, a.ldwb = X.data_ptr(), D, W1.data_ptr(), D, W2.data_ptr(), H
        a.M, a.H, a.mode = M, H, mode
        a.bias_a, a.bias_b, a.resid, a.ldr = b1.data_ptr(), b2.data_ptr(), resid.data_ptr(), D
        a.out, a.ldo = out.data_ptr(), D
        a.save_grad, a.ldsg, a.save_act, a.ldsa = gs.data_ptr(), H, a_.data_ptr(), H
        runs.append((n, mode, fn, a))
st = torch.cuda.current_stream().cuda_stream
best = {}
for rnd in range(4):                                   # interleaved rounds, minimum per variant (the clock drifts over a call)
    for n, mode, fn, a in runs:
        us = timeit(lambda: fn(C.byref(a), C.c_void_p(st)), 20)
        best[(n, mode)] = min(best.get((n, mode), 1e9), us)
for (n, mode), us in best.items():
    print(f"abl {n:4d} ({names.get(n, '?'):24s}) mode {mode}: {us:6.1f} us", flush=True)

# ---- unit sizes: 256 full units (128 rows, 8 computing waves) against 256 units of 64 / 48 / 32 rows (libmlp_uw<N>.so: N computing waves)
Mbig = 128 * 256
Xb = torch.randn(Mbig, D, generator=g).to(bf16).to(DEV); rb = torch.randn(Mbig, D, generator=g).to(DEV); ob = torch.empty(Mbig, D, device=DEV)
gb = torch.empty(Mbig, H, device=DEV, dtype=bf16); ab = torch.empty(Mbig, H, device=DEV, dtype=bf16)
cases = [(None, 8)] + [(os.path.join(here, "libmlp_uw%d.so" % n), n) for n in (4, 3, 2) if os.path.isfile(os.path.join(here, "libmlp_uw%d.so" % n))]
for mode in (_lib.MLP_FWD, _lib.MLP_FWD_SAVE, _lib.MLP_BWD):
    line = []
    for path, uw in cases:
        h = _lib.lib() if path is None else C.CDLL(path)
        fn = h.lafs_mlp_fused
        fn.argtypes = [C.POINTER(_lib.MlpArgs), C.c_void_p]; fn.restype = C.c_int
        a = _lib.MlpArgs()
        a.X, a.ldx, a.Wa, a.ldwa, a.Wb, a.ldwb = Xb.data_ptr(), D, W1.data_ptr(), D, W2.data_ptr(), H
        a.M, a.H, a.mode = 16 * uw * 256, H, mode
        a.bias_a, a.bias_b, a.resid, a.ldr = b1.data_ptr(), b2.data_ptr(), rb.data_ptr(), D
        a.out, a.ldo = (ob.data_ptr() if mode != _lib.MLP_BWD else Xb.data_ptr()), D
        a.save_grad, a.ldsg, a.save_act, a.ldsa = gb.data_ptr(), H, ab.data_ptr(), H
        us = min(timeit(lambda: fn(C.byref(a), C.c_void_p(st)), 20) for _ in range(3))
        line.append(f"{uw} waves x 256 units: {us:6.1f} us")
    print(f"mode {mode}: " + "   ".join(line), flush=True)
