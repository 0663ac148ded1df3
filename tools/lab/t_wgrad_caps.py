"""GPU box: lafs_wgrad_group at the ViT-S / Part-fViT block shapes under every workgroup cap the engines use, against fp32 torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import ops
dev = "cuda"
torch.manual_seed(0)
shapes = {"ViT-S": (44160, [(384, 1536), (1536, 384), (384, 384), (1152, 384)]),
          "Part-fViT student": (44160, [(768, 2048), (2048, 768), (768, 704), (2112, 768)]),
          "Part-fViT 25216": (25216, [(768, 2048), (2048, 768), (768, 704), (2112, 768)]),
          "Part-fViT 18944": (18944, [(768, 2048), (2048, 768), (768, 704), (2112, 768)])}
worst = 0.0
for name, (M, dims) in shapes.items():
    pairs = [(torch.randn(M, a, device=dev).to(torch.bfloat16), torch.randn(M, b, device=dev).to(torch.bfloat16)) for a, b in dims]
    refs = [x.float().t() @ y.float() for x, y in pairs]
    cref = [x.float().sum(0) for x, _ in pairs]
    for cap in (0, 200, 160, 96, 248):
        for acc in (False, True):
            Cs = [torch.full((a, b), 0.5 if acc else 7.0, device=dev) for a, b in dims]
            cs = [torch.zeros(a, device=dev) for a, _ in dims]
            ops.wgrad_group([(x, y, c, acc, s) for (x, y), c, s in zip(pairs, Cs, cs)], max_workgroups=cap)
            torch.cuda.synchronize()
            for c, r, s, sr in zip(Cs, refs, cs, cref):
                e = float(((c - (0.5 if acc else 0.0)) - r).norm() / r.norm()); es = float((s - sr).norm() / sr.norm())
                worst = max(worst, e, es)
                if e > 2e-3 or es > 2e-3:
                    print("MISMATCH", name, "cap", cap, "acc", acc, tuple(c.shape), e, es)
print("worst rel-L2", worst)
