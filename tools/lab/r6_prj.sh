#!/bin/bash
# round 6 lab: the attention projection in front of the fused MLP (LAFS_OPT_MLP_FUSED bits 128 teacher / 256 student): step parity
# tests with both on, and a same-box A/B of the headline step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( LAFS_MLP_FUSED=463 timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_modules.py tests/test_gpu_composition.py -x -q -m gpu 2>&1 | tail -5
  ENVS='LAFS_MLP_FUSED=79|LAFS_MLP_FUSED=207|LAFS_MLP_FUSED=463|LAFS_MLP_FUSED=335' bash tools/lab/ab_env_headline.sh ) > gpurun_out/r6_prj.txt 2>&1
cat gpurun_out/r6_prj.txt
