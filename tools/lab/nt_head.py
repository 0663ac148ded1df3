#!/usr/bin/env python3
"""Lab: the DINO head's small-M GEMMs (640 / 128 rows) on 64-deep 2-stage (library choice) vs 32-deep 3-stage rings (debug flag 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lafs_cvpr2024_amd import _lib, ops
sys.argv = ["x", "none"]
exec(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_kernels.py")).read().split("SHAPES = [")[0])
for flag in (0, 2, 0, 2):
    _lib.lib().lafs_debug_set(flag)
    print("debug flags", flag)
    for M in (640, 128):
        nt(M, 2048, 384, _lib.EPI_BF16_GELU, "mlp.0"); nt(M, 2048, 2048, _lib.EPI_BF16_GELU, "mlp.2"); nt(M, 256, 2048, _lib.EPI_F32, "mlp.4")
        nt(M, 2048, 256, _lib.EPI_DGELU_BF16, "mlp.4 dgrad"); nt(M, 2048, 2048, _lib.EPI_DGELU_BF16, "mlp.2 dgrad"); nt(M, 384, 2048, _lib.EPI_F32, "mlp.0 dgrad")
_lib.lib().lafs_debug_set(0)
