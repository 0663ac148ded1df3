"""GPU box: LayerNorm backward at D = 384 / 768, with and without the dropout mask and the DropPath scale (what mynet runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import ops
dev = "cuda"
T = 44160
def timeit(fn, n=200):
    for _ in range(n): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
lens = [197] * 128 + [37] * 512
row2seq = torch.repeat_interleave(torch.arange(640, device=dev, dtype=torch.int32), torch.tensor(lens, device=dev))
scale = (torch.rand(640, device=dev) > 0.1).float() / 0.9
for D in (384, 768):
    x = torch.randn(T, D, device=dev); dy = torch.randn(T, D, device=dev).to(torch.bfloat16); dyf = torch.randn(T, D, device=dev)
    stats = torch.stack([x.mean(1), x.var(1, unbiased=False).add(1e-6).rsqrt()], 1).contiguous()
    gamma = torch.ones(D, device=dev); g = torch.zeros(T, D, device=dev); dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
    gb = torch.empty(T, D, device=dev, dtype=torch.bfloat16)
    byt = T * D * (2 + 4 + 4 + 4)
    for name, kw in (("plain", {}), ("gb_out", dict(gb_out=gb)), ("gb_out+scale", dict(gb_out=gb, seq_scale=scale, row2seq=row2seq)),
                     ("gb_out+scale+dropout", dict(gb_out=gb, seq_scale=scale, row2seq=row2seq, drop_p=0.1, drop_seed=5)),
                     ("dropout only", dict(drop_p=0.1, drop_seed=5))):
        t = timeit(lambda: ops.layernorm_bwd(dy, x, stats, gamma, g, dg, db, accumulate=True, **kw))
        print(f"D={D} {name:24s} {t:7.1f} us  {byt / t / 1e6:6.2f} TB/s (of the plain form's bytes)")
    t = timeit(lambda: ops.layernorm_bwd(dyf, x, stats, gamma, g, dg, db, accumulate=True))
    print(f"D={D} f32 dy                   {t:7.1f} us")
