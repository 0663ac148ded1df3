#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b14; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "big_tiles and 5" -p no:cacheprovider 2>&1 | tail -3
ENVS="LAFS_NT_BIG=0|LAFS_NT_BIG=1|LAFS_NT_BIG=2" WHICH=mynet bash tools/lab/ab_env_mynet.sh 2>&1 | tee $O/mynet.txt
ENVS="LAFS_NT_BIG=0|LAFS_NT_BIG=1|LAFS_NT_BIG=2" WHICH=finetune bash tools/lab/ab_env_mynet.sh 2>&1 | tee $O/finetune.txt
timeout 600 python tools/lab/t_big_ab.py 2>&1 | grep "^M=25216" | tee $O/big_ab.txt
