#!/usr/bin/env python3
"""Lab: the long-K GEMMs of the ViT-S trunk + 4096^3 on the tiled kernel (A/B two library builds: LAFS_USE_ABLATE_LIB=1 loads the other .so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops
sys.argv = ["x", "none"]
exec(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_kernels.py")).read().split("SHAPES = [")[0])
print("LAFS_USE_ABLATE_LIB =", os.environ.get("LAFS_USE_ABLATE_LIB"))
for rep in range(2):
    for T_ in (44160, 25216):
        nt(T_, 384, 1536, _lib.EPI_RESID_F32, "fc2 fwd"); nt(T_, 384, 1536, _lib.EPI_BF16, "fc1 dgrad"); nt(T_, 384, 1152, _lib.EPI_BF16, "qkv dgrad")
    nt(4096, 4096, 4096, _lib.EPI_BF16, "4096^3")
    nt(25216, 768, 2048, _lib.EPI_BF16, "ViT-B fc1 dgrad"); nt(25216, 2048, 768, _lib.EPI_BF16_GELU, "ViT-B fc1")
