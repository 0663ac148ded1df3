#!/usr/bin/env python3
"""Lab: the long-K GEMMs of the ViT-S trunk on the tiled kernel's other tile shapes (lafs_debug_set: 0 library choice = 128x128 tiles
with 64-deep stages; 4 = 256x128 / 64-deep, one workgroup per CU; 6 = 256x128 / 32-deep 3-stage ring, two per CU; 2 = 128x128 / 32-deep)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops
sys.argv = ["x", "none"]
exec(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_kernels.py")).read().split("SHAPES = [")[0])
for flag in (0, 4, 6, 2, 0):
    _lib.lib().lafs_debug_set(flag)
    print("debug flags", flag)
    for T_ in (44160, 25216):
        nt(T_, 384, 1536, _lib.EPI_RESID_F32, "fc2 fwd"); nt(T_, 384, 1536, _lib.EPI_BF16, "fc1 dgrad"); nt(T_, 384, 1152, _lib.EPI_BF16, "qkv dgrad")
_lib.lib().lafs_debug_set(0)
