#!/bin/bash
# A/B on ONE box: whole-step bench under different environments, e.g.  ENVS="LAFS_KRES=0|LAFS_KRES=7|LAFS_KRES_GRID=1024" bash tools/lab/ab_env.sh
IFS='|' read -ra LIST <<< "${ENVS:-X=0}"
for rep in 1 2; do
for e in "${LIST[@]}"; do
  echo "=== $e"
  env $e timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 < /dev/null | grep -o '"ms_per_step": [0-9.]*'
done
done
