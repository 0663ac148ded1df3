#!/usr/bin/env python3
"""Lab (ablation build: LAFS_USE_ABLATE_LIB=1): persistent workgroups with next-tile prefetch (debug flag 8388608) on the DINO head's
last layer (K = 256: four stages per tile, 3910 tiles) and the long-K trunk shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops
sys.argv = ["x", "none"]
exec(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_kernels.py")).read().split("SHAPES = [")[0])
for flag in (0, 8388608, 0, 8388608):
    _lib.lib().lafs_debug_set(flag)
    print("debug flags", flag)
    nt(640, 100096, 256, _lib.EPI_F32, "last layer (student)"); nt(128, 100096, 256, _lib.EPI_F32, "last layer (teacher)")
    nt(44160, 384, 1536, _lib.EPI_RESID_F32, "fc2 fwd"); nt(44160, 384, 1536, _lib.EPI_BF16, "fc1 dgrad")
_lib.lib().lafs_debug_set(0)
