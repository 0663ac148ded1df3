"""Lab: the fused MLP against the two-launch path over many (rows, hidden) shapes -- every output bit for bit (rows >= 2048: GEMM 1 of the
two-launch path on the K-resident kernel), all three modes + the LayerNorm prologue (rows >= 4096: the two-rows-per-wave LayerNorm kernel).
usage: python tools/lab/fuzz_mlp_fused.py [n_cases] [seed]"""
import os
import random
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops
from lafs_cvpr2024_amd.ops import _p, call

DEV, bf16, D = "cuda", torch.bfloat16, 384
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    M = rng.choice([2048, 2049, 2048 + 16 * rng.randint(1, 40) + rng.randint(0, 15), 4096 + rng.randint(0, 3000), 128 * rng.randint(17, 70),
                    128 * 256 + rng.randint(1, 900), 128 * 300 + rng.randint(0, 127)])
    H = 64 * rng.randint(2, 24)
    g = torch.Generator().manual_seed(1000 + case)
    X = torch.randn(M, D, generator=g).to(bf16).to(DEV)
    W1 = (torch.randn(H, D, generator=g) * 0.05).to(bf16).to(DEV); W2 = (torch.randn(D, H, generator=g) * 0.03).to(bf16).to(DEV)
    b1, b2 = (torch.randn(H, generator=g) * 0.1).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV)
    resid = torch.randn(M, D, generator=g).to(DEV)
    nseq = rng.randint(1, 9)
    row2seq = (torch.arange(M) * nseq // M).int().to(DEV)
    sc = torch.tensor([0.0 if rng.random() < 0.2 else 1.0 / 0.9 for _ in range(nseq)]).to(DEV)
    kw = dict(bias_a=b1, bias_b=b2, resid=resid, seq_scale=sc, row2seq=row2seq)
    gs, a_s = ops.gemm_nt(X, W1, _lib.EPI_BF16_GELU, bias=b1, act=1)
    y_s = ops.gemm_nt(a_s, W2, _lib.EPI_RESID_F32, bias=b2, resid=resid, seq_scale=sc, row2seq=row2seq)
    y = ops.mlp_fused(X, W1, W2, _lib.MLP_FWD, **kw)[0]
    y2, g2, a2 = ops.mlp_fused(X, W1, W2, _lib.MLP_FWD_SAVE, **kw)
    ok = torch.equal(y, y_s) and torch.equal(y2, y_s) and torch.equal(g2, gs) and torch.equal(a2, a_s)
    dY = torch.randn(M, D, generator=g).to(bf16).to(DEV)
    W2t, W1t = W2.t().contiguous(), W1.t().contiguous()
    du_ref = ops.gemm_nt(dY, W2t, _lib.EPI_DGELU_BF16, aux=gs, act=1)
    dx_ref = ops.gemm_nt(du_ref, W1t, _lib.EPI_BF16)
    dx, _, du = ops.mlp_fused(dY, W2t, W1t, _lib.MLP_BWD, save_grad=gs)
    ok = ok and torch.equal(du, du_ref) and torch.equal(dx, dx_ref)
    if M >= 4096:
        gam, bet = (1.0 + 0.2 * torch.randn(D, generator=g)).to(DEV), (0.1 * torch.randn(D, generator=g)).to(DEV)
        h = torch.empty(M, D, device=DEV, dtype=bf16); st = torch.empty(M, 2, device=DEV)
        call("lafs_layernorm_fwd", _p(resid), D, _p(gam), _p(bet), 1e-6, _p(h), D, None, 0, _p(st), M, D)
        yr = ops.mlp_fused(h, W1, W2, _lib.MLP_FWD_SAVE, **kw)
        h2 = torch.empty_like(h); st2 = torch.empty_like(st)
        yl = ops.mlp_fused(None, W1, W2, _lib.MLP_FWD_SAVE, ln=(gam, bet, 1e-6), ln_stats=st2, ln_out=h2, **kw)
        ok = ok and torch.equal(h2, h) and torch.equal(st2, st) and all(torch.equal(a, b) for a, b in zip(yl, yr))
    torch.cuda.synchronize()
    if not ok:
        bad += 1
    print(f"case {case:3d}: M = {M:6d} H = {H:4d} nseq = {nseq}: {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{n_cases - bad} of {n_cases} cases bit-identical")
sys.exit(1 if bad else 0)
