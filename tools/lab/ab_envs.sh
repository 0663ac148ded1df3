#!/bin/bash
# one box: bench.py under several environment settings, two rounds.  usage: tools/lab/ab_envs.sh "A=1" "A=2" ...
cd $GRAFT_REPO_ROOT
for i in 1 2; do for e in "$@"; do
  echo "$e: $(env $e python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
done; done
