#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -s -k "layernorm_backward_in_its_epilogue" 2>&1 | grep -E "mlp-bwd|passed|failed|Error|assert" | tail -4 | tee gpurun_out/r6_b7_test.txt
ENVS='X=0|LAFS_WGRAD_WG=160|LAFS_WGRAD_WG=240|LAFS_WGRAD_WG=0|LAFS_ROW_CHAINS=4|LAFS_TEACHER_SERIAL=1|LAFS_KRES_MIN_ITEMS=2|LAFS_KRES_MIN_ITEMS=8' bash tools/lab/ab_env_headline.sh 2>&1 | tee gpurun_out/r6_b7_ab.txt
