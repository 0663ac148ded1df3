#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k attention -p no:cacheprovider 2>&1 | tail -2
for v in prevattn "" prevattn ""; do echo "== variant '$v' $(LAFS_LIB_VARIANT=$v timeout 300 python tools/bench_kernels.py attn 2>&1 | grep ' x ' | tr '\n' ' ')"; done
ENVS="LAFS_LIB_VARIANT=prevattn|LAFS_LIB_VARIANT=" bash tools/lab/ab_env.sh 2>&1
