#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python tools/lab/t_mlp_fused.py 30 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_b2_time.txt
