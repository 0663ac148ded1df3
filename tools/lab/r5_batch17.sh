#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b17; mkdir -p $O; cd $R
ENVS="LAFS_LIB_VARIANT=attn_old|LAFS_LIB_VARIANT=" bash tools/lab/ab_env.sh 2>&1 | tee $O/c2.txt
ENVS="LAFS_LIB_VARIANT=attn_old|LAFS_LIB_VARIANT=" WHICH=mynet bash tools/lab/ab_env_mynet.sh 2>&1 | tee $O/mynet.txt
ENVS="LAFS_LIB_VARIANT=attn_old|LAFS_LIB_VARIANT=" WHICH=finetune bash tools/lab/ab_env_mynet.sh 2>&1 | tee $O/finetune.txt
