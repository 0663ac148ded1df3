#!/bin/bash
# A/B on ONE box: whole-step bench with the K-resident GEMM routed for different epilogue sets (LAFS_KRES bit mask:
# 1 plain, 2 GELU, 4 residual, 8 GELU'; 0 = tiled kernel everywhere).  EXTRA_ENV e.g. LAFS_SINGLE_STREAM=1
for m in ${MASKS:-0 15 7 6 0 15}; do
  echo "=== LAFS_KRES=$m $EXTRA_ENV"
  env LAFS_KRES=$m $EXTRA_ENV timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
