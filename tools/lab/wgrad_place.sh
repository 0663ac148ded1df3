#!/bin/bash
# GPU box: the grouped weight gradient with / without bias-gradient column sums: sustained time and fabric reads (tools/lab/lab_wgrad)
cd $GRAFT_REPO_ROOT; export LAB_SHORT=1
for cfg in "LAB_ALIGN=2097152" "LAB_ALIGN=2097152 LAB_ACC=1"; do
  echo "== $cfg"; env $cfg LAB_ITERS=4000 tools/lab/lab_wgrad 2>/dev/null
  env $cfg bash tools/lab/pmc_mem.sh tools/lab/lab_wgrad 2>&1 | grep "wgrad_kernel"
done
