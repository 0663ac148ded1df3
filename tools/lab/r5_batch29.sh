#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b29; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_finetune.py tests/test_gpu_modules.py -q -x -p no:cacheprovider 2>&1 | tail -3
for v in prevhash ""; do echo "== variant '$v'"; LAFS_LIB_VARIANT=$v timeout 900 python tools/lab/t_big_ab.py 2>&1 | grep "^M=" | grep "GELU pair\|dgelu\|fc2 fwd" | cut -c1-330; done
ENVS="LAFS_LIB_VARIANT=prevhash|LAFS_LIB_VARIANT=" WHICH=mynet bash tools/lab/ab_env_mynet.sh 2>&1
ENVS="LAFS_LIB_VARIANT=prevhash|LAFS_LIB_VARIANT=" WHICH=finetune bash tools/lab/ab_env_mynet.sh 2>&1
