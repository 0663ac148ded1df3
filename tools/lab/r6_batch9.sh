#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for rep in 1 2; do
  echo "== new"; python tools/bench_kernels.py attn 2>&1 | grep " x "
  echo "== prev"; LAFS_LIB_VARIANT=prevattn python tools/bench_kernels.py attn 2>&1 | grep " x "
done | tee gpurun_out/r6_b9_attn.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k attention 2>&1 | tail -2
ENVS='X=0|LAFS_LIB_VARIANT=prevattn' bash tools/lab/ab_env_headline.sh 2>&1 | tee gpurun_out/r6_b9_ab.txt
