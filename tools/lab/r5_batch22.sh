#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b22; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm" -p no:cacheprovider 2>&1 | tail -2
ENVS="LAFS_LIB_VARIANT=oldrule|LAFS_LIB_VARIANT=|LAFS_NT_BIG=0" WHICH=mynet bash tools/lab/ab_env_mynet.sh 2>&1 | tee $O/mynet.txt
ENVS="LAFS_LIB_VARIANT=oldrule|LAFS_LIB_VARIANT=|LAFS_NT_BIG=0" WHICH=finetune bash tools/lab/ab_env_mynet.sh 2>&1 | tee $O/finetune.txt
