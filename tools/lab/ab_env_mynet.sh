#!/bin/bash
# A/B on ONE box: the mynet pair (bench.py --extras-only mynet) under different environments, e.g. ENVS="LAFS_ROW_CHAINS=2|LAFS_ROW_CHAINS=1"
IFS='|' read -ra LIST <<< "${ENVS:-X=0}"
for rep in 1 2; do
for e in "${LIST[@]}"; do
  echo "=== $e"
  env $e timeout 300 python bench.py --extras-only ${WHICH:-mynet} --no-roofline 2>&1 < /dev/null | grep -o '"ms_per_step": [0-9.]*'
done
done
