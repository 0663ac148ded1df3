#!/bin/bash
IFS='|' read -ra LIST <<< "${ENVS:-X=0}"
for rep in 1 2 3; do
for e in "${LIST[@]}"; do
  echo "=== $e"
  env $e timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras --no-roofline 2>&1 < /dev/null | grep -o '"ms_per_step": [0-9.]*'
done
done
