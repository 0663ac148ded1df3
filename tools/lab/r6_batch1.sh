#!/bin/bash
# round 6, call 1: the fused MLP kernel -- bit-identity test, then its timing against the two launches
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_mlp" 2>&1 | tail -15 > gpurun_out/r6_b1_test.txt
cat gpurun_out/r6_b1_test.txt
timeout 300 python tools/lab/t_mlp_fused.py 40 2>&1 | tee gpurun_out/r6_b1_time.txt
