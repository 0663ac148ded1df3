#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b15; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k attention -p no:cacheprovider 2>&1 | tail -5
for v in attn_old ""; do
  echo "== variant '$v'"
  LAFS_LIB_VARIANT=$v timeout 300 python tools/bench_kernels.py attn 2>&1 | grep " x "
  LAFS_LIB_VARIANT=$v timeout 300 python tools/bench_kernels.py attn 2>&1 | grep " x "
done
