"""Lab: lafs_layernorm_bwd, two rows per wave (product, ln_bwd2_kernel) against one row per wave (tools/lab/libln_bwd1.so: the same
source with -DLAFS_LAB_LN_BWD1), ViT-S row counts, HIP events, interleaved; and the difference of their results.
usage: python tools/lab/t_ln_bwd.py [reps]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from lafs_cvpr2024_amd import _lib

DEV, bf16 = "cuda", torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
D = 384
new = _lib.lib()
old = C.CDLL(os.path.join(ROOT, "tools", "lab", "libln_bwd1.so"))
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def run(lib, dy, x, st, gam, g, gb, sc, r2s, part, M):
    rc = lib.lafs_layernorm_bwd(P(dy), D, None, 0, P(x), D, P(st), P(gam), P(g), D, 1, P(gb), D, P(sc), P(r2s), None, None, M, D,
                                C.c_float(0.0), 0, None, 0, P(part), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc


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


for M in (25216, 12608, 18944, 44160):
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(M, D, generator=g) * 1.3 + 0.2).to(DEV)
    gam = (1.0 + 0.2 * torch.randn(D, generator=g)).to(DEV); bet = torch.zeros(D, device=DEV)
    h = torch.empty(M, D, device=DEV, dtype=bf16); st = torch.empty(M, 2, device=DEV)
    new.lafs_layernorm_fwd(P(x), D, P(gam), P(bet), C.c_float(1e-6), P(h), D, None, 0, P(st), M, D, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    dy = torch.randn(M, D, generator=g).to(bf16).to(DEV)
    g0 = torch.randn(M, D, generator=g).to(DEV)
    r2s = (torch.arange(M) * 64 // M).int().to(DEV); sc = torch.full((64,), 1.0 / 0.9, device=DEV)
    nparts = int(new.lafs_layernorm_bwd_parts(M, D))
    outs = {}
    for name, lib in (("one", old), ("two", new)):
        gg = g0.clone(); gb = torch.empty(M, D, device=DEV, dtype=bf16); part = torch.zeros(nparts, 2, D, device=DEV)
        run(lib, dy, x, st, gam, gg, gb, sc, r2s, part, M)
        outs[name] = (gg, gb.float(), part.double().sum(0))
    dif = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(outs["two"], outs["one"])]
    gg = g0.clone(); gb = torch.empty(M, D, device=DEV, dtype=bf16); part = torch.zeros(nparts, 2, D, device=DEV)
    res = {}
    for rnd in range(3):
        for name, lib in (("one", old), ("two", new)):
            res.setdefault(name, []).append(timeit(lambda: run(lib, dy, x, st, gam, gg, gb, sc, r2s, part, M)))
    byt = M * D * 16.0
    print(f"M={M:6d} parts={nparts}  one row/wave {min(res['one']):6.1f} us ({byt / min(res['one']) / 1e6:.2f} TB/s)   two rows/wave "
          f"{min(res['two']):6.1f} us ({byt / min(res['two']) / 1e6:.2f} TB/s)   max rel diff g / gb / (dgamma, dbeta): "
          + " ".join(f"{d:.1e}" for d in dif), flush=True)
