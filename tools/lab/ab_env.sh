#!/bin/bash
# A/B on ONE box: bench.py with and without an environment setting, interleaved.  usage: tools/lab/ab_env.sh VAR=VALUE
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo "base: $(python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
  echo "$1: $(env $1 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
done
