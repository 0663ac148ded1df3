#!/bin/bash
# round 6, call 3: fused MLP in the LAFS step -- composition parity with it on, then the step A/B over the mask
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
LAFS_MLP_FUSED=7 timeout 900 python -m pytest tests/test_gpu_composition.py tests/test_gpu_step.py -x -q -m gpu -k "composition or f17 or F17 or reference_step" 2>&1 | tail -5 | tee gpurun_out/r6_b3_test.txt
ENVS='LAFS_MLP_FUSED=0|LAFS_MLP_FUSED=1|LAFS_MLP_FUSED=2|LAFS_MLP_FUSED=4|LAFS_MLP_FUSED=7|LAFS_MLP_FUSED=7 LAFS_ROW_CHAINS=1|LAFS_MLP_FUSED=0 LAFS_ROW_CHAINS=1' bash tools/lab/ab_env_headline.sh 2>&1 | tee gpurun_out/r6_b3_ab.txt
