#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_mlp" 2>&1 | tail -2
python tools/lab/t_mlp_stamps.py 2>&1 | grep -v amdgpu.ids | grep "M=25216" | cut -c1-400
ENVS='X=0|LAFS_LIB_VARIANT=prev' bash tools/lab/ab_env_headline.sh 2>&1
