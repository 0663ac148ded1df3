#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b20; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_finetune.py -q -x -s -p no:cacheprovider -k "stage_by_stage" > $O/finetune.log 2>&1; echo "rc=$?"; grep "grad-gate\|Error\|passed\|failed" $O/finetune.log | cut -c1-400
