#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_step.py tests/test_gpu_modules.py tests/test_gpu_composition.py tests/test_gpu_finetune.py -x -q -m gpu 2>&1 | tail -4
sed -i 's/for lib in "" oldln; do/for lib in ""; do/' tools/lab/chains_probe.sh
bash tools/lab/chains_probe.sh 2>&1 | tee gpurun_out/r6_b12_probe.txt
ENVS='X=0' bash tools/lab/ab_env_headline.sh
