#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -s -k "fused_mlp" 2>&1 | grep -E "mlp-bwd|passed|failed|Error|assert" | tail -8 | tee gpurun_out/r6_b6_test.txt
LAFS_MLP_FUSED=31 timeout 900 python -m pytest tests/test_gpu_composition.py tests/test_gpu_step.py -x -q -m gpu -k "composition or f17 or F17 or reference_step" 2>&1 | tail -5 | tee -a gpurun_out/r6_b6_test.txt
ENVS='LAFS_MLP_FUSED=0|LAFS_MLP_FUSED=15|LAFS_MLP_FUSED=31' bash tools/lab/ab_env_headline.sh 2>&1 | tee gpurun_out/r6_b6_ab.txt
