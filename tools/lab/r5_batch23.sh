#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b23; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm" -p no:cacheprovider 2>&1 | tail -2
ENVS="LAFS_LIB_VARIANT=tiledold|LAFS_LIB_VARIANT=" bash tools/lab/ab_env.sh 2>&1 | tee $O/c2.txt
