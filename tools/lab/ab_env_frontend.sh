#!/bin/bash
# A/B on ONE box: the LAFS step with the landmark front-end in it (bench.py --frontend) under different environments
IFS='|' read -ra LIST <<< "${ENVS:-X=0}"
for rep in 1 2; do
for e in "${LIST[@]}"; do
  echo "=== $e"
  env $e timeout 300 python bench.py --frontend --steps 30 --warmup 6 --no-cpu-baseline --no-extras --no-roofline 2>&1 < /dev/null | grep -o '"ms_per_step": [0-9.]*'
done
done
