#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export LAFS_MLP_FUSED=15
bash tools/profile_serial.sh 2>&1 | tail -60 | tee gpurun_out/r6_b5_serial.txt
