#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_mlp" 2>&1 | tail -3
python tools/lab/t_mlp_stamps.py 2>&1 | grep -v amdgpu.ids | grep "M=25216" | tee gpurun_out/r6_b10_stamps.txt
timeout 300 python tools/lab/t_mlp_fused.py 30 2>&1 | grep "^M=" | tee gpurun_out/r6_b10_time.txt
ENVS='X=0' bash tools/lab/ab_env_headline.sh 2>&1 | tee gpurun_out/r6_b10_ab.txt
