#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b25; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_finetune.py tests/test_gpu_kernels.py -q -x -p no:cacheprovider 2>&1 | tail -3
ENVS="LAFS_LIB_VARIANT=prev|LAFS_LIB_VARIANT=" bash tools/lab/ab_env.sh 2>&1 | tee $O/c2.txt
for v in prev ""; do echo "== frontend variant '$v'"; LAFS_LIB_VARIANT=$v timeout 300 python bench.py --frontend --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | grep -o '"ms_per_step": [0-9.]*'; done
