#!/bin/bash
# A/B on ONE box: bench.py with liblafs_hip.so (new) against whatever was copied to liblafs_hip_ablate.so (old), interleaved.
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo "new: $(python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
  echo "old: $(LAFS_USE_ABLATE_LIB=1 python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
done
