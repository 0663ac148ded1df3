#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "eight_rank" 2>&1 | tail -3 ) 2>&1 | tee gpurun_out/r6_b8_test.txt
LAFS_MLP_FUSED=47 timeout 900 python -m pytest tests/test_gpu_composition.py -x -q -m gpu -k "composition" 2>&1 | tail -3 | tee -a gpurun_out/r6_b8_test.txt
ENVS='LAFS_MLP_FUSED=15|LAFS_MLP_FUSED=47' bash tools/lab/ab_env_headline.sh 2>&1 | tee gpurun_out/r6_b8_ab.txt
