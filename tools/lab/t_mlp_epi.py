"""Lab: what the fused MLP's optional prologue / epilogue cost per launch at the C2 row counts (HIP events, interleaved, min of rounds):
plain | LayerNorm 2 prologue | + the next block's LayerNorm 1 in the epilogue, forward-only and saving form; and the LayerNorm launch
each replaces.  usage: python tools/lab/t_mlp_epi.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops
from lafs_cvpr2024_amd.ops import _p, call

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


for M in (25216, 18944, 12608):
    g = torch.Generator().manual_seed(1)
    W1 = (torch.randn(H, D, generator=g) * 0.05).to(bf16).to(DEV); W2 = (torch.randn(D, H, generator=g) * 0.03).to(bf16).to(DEV)
    b1, b2 = torch.randn(H, generator=g).to(DEV) * 0.1, torch.randn(D, generator=g).to(DEV) * 0.1
    gam, bet = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    resid = torch.randn(M, D, generator=g).to(DEV)
    X = torch.empty(M, D, device=DEV, dtype=bf16); st = torch.empty(M, 2, device=DEV)
    out = torch.empty(M, D, device=DEV)
    a = torch.empty(M, H, device=DEV, dtype=bf16); gs = torch.empty(M, H, device=DEV, dtype=bf16)
    hn = torch.empty(M, D, device=DEV, dtype=bf16); sn = torch.empty(M, 2, device=DEV); h2 = torch.empty(M, D, device=DEV, dtype=bf16)
    row2seq = (torch.arange(M) * 64 // M).int().to(DEV); sc = torch.full((64,), 1.0 / 0.9, device=DEV)
    kw = dict(bias_a=b1, bias_b=b2, resid=resid, seq_scale=sc, row2seq=row2seq, out=out)
    ln = lambda: call("lafs_layernorm_fwd", _p(resid), D, _p(gam), _p(bet), 1e-6, _p(X), D, None, 0, _p(st), M, D)
    ln()
    sv = dict(save_grad=gs, save_act=a)
    o = torch.randn(M, D, generator=g).to(bf16).to(DEV); Wp = (torch.randn(D, D, generator=g) * 0.05).to(bf16).to(DEV)
    x0 = torch.randn(M, D, generator=g).to(DEV)
    kwp = dict(kw); kwp["resid"] = resid                 # (resid is written by the projection prologue)
    fns = {
        "proj launch": lambda: ops.gemm_nt(o, Wp, _lib.EPI_RESID_F32, bias=b2, out=resid, resid=x0, seq_scale=sc, row2seq=row2seq),
        "LN launch": ln,
        "fwd plain": lambda: ops.mlp_fused(X, W1, W2, _lib.MLP_FWD, **kw),
        "fwd +LN2": lambda: ops.mlp_fused(None, W1, W2, _lib.MLP_FWD, ln=(gam, bet, 1e-6), **kw),
        "fwd +LN2 +nextLN1": lambda: ops.mlp_fused(None, W1, W2, _lib.MLP_FWD, ln=(gam, bet, 1e-6), next_ln=(gam, bet, 1e-6, hn, None), **kw),
        "save plain": lambda: ops.mlp_fused(X, W1, W2, _lib.MLP_FWD_SAVE, **sv, **kw),
        "save +LN2": lambda: ops.mlp_fused(None, W1, W2, _lib.MLP_FWD_SAVE, ln=(gam, bet, 1e-6), ln_stats=st, ln_out=h2, **sv, **kw),
        "save +LN2 +nextLN1": lambda: ops.mlp_fused(None, W1, W2, _lib.MLP_FWD_SAVE, ln=(gam, bet, 1e-6), ln_stats=st, ln_out=h2,
                                                    next_ln=(gam, bet, 1e-6, hn, sn), **sv, **kw),
        "fwd proj+LN2+nextLN1": lambda: ops.mlp_fused(None, W1, W2, _lib.MLP_FWD, ln=(gam, bet, 1e-6), next_ln=(gam, bet, 1e-6, hn, None),
                                                      proj=(o, Wp, b2, x0, sc), **kw),
        "save proj+LN2+nextLN1": lambda: ops.mlp_fused(None, W1, W2, _lib.MLP_FWD_SAVE, ln=(gam, bet, 1e-6), ln_stats=st, ln_out=h2,
                                                       next_ln=(gam, bet, 1e-6, hn, sn), proj=(o, Wp, b2, x0, sc), **sv, **kw),
    }
    res = {}
    for rnd in range(3):
        for name, fn in fns.items():
            res.setdefault(name, []).append(timeit(fn))
    print(f"M={M}: " + "  ".join(f"{k} {min(v):6.1f}" for k, v in res.items()), flush=True)
