#!/usr/bin/env python3
"""Lab: the K = 384 GEMMs of the ViT-S trunk on the K-resident kernel (A/B two library builds: LAFS_USE_ABLATE_LIB=1 loads the other .so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops
sys.argv = ["x", "none"]
exec(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_kernels.py")).read().split("SHAPES = [")[0])
print("LAFS_USE_ABLATE_LIB =", os.environ.get("LAFS_USE_ABLATE_LIB"))
for rep in range(2):
    for T_ in (44160, 25216):
        nt(T_, 1152, 384, _lib.EPI_BF16, "qkv"); nt(T_, 384, 384, _lib.EPI_RESID_F32, "proj"); nt(T_, 1536, 384, _lib.EPI_BF16_GELU, "fc1")
        nt(T_, 1536, 384, _lib.EPI_DGELU_BF16, "fc2 dgrad"); nt(T_, 384, 384, _lib.EPI_BF16, "proj dgrad")
