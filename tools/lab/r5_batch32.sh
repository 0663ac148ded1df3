#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for e in "X=0" "LAFS_ROW_CHAINS=1" "LAFS_WGRAD_WG=160" "LAFS_WGRAD_WG=240" "LAFS_NT_WIDE=0" "LAFS_NT_TALL=0"; do echo "=== $e"; env $e timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-roofline 2>&1 < /dev/null | grep -o '"ms_per_step": [0-9.]*'; done; done
