#!/bin/bash
# round 6 lab: the next block's LayerNorm 1 in the fused MLP's epilogue (LAFS_OPT_MLP_FUSED bit 64) -- its test, the step parity
# tests with it on, and a same-box A/B of the headline step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "next_blocks_layernorm or fused_mlp" 2>&1 | tail -5
  LAFS_MLP_FUSED=79 timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_modules.py tests/test_gpu_composition.py -x -q -m gpu 2>&1 | tail -5
  ENVS='LAFS_MLP_FUSED=15|LAFS_MLP_FUSED=79' bash tools/lab/ab_env_headline.sh ) > gpurun_out/r6_nextln.txt 2>&1
cat gpurun_out/r6_nextln.txt
