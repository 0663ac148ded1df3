#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b3; mkdir -p $O; cd $R
timeout 600 tools/lab/lab_big > $O/lab_big.log 2>&1; echo "lab_big rc=$?"; cat $O/lab_big.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -p no:cacheprovider -k "big_tiles or k_resident or tall or wgrad or layernorm" > $O/kern.log 2>&1; echo "kern rc=$?"; tail -5 $O/kern.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_finetune.py -x -q -s -p no:cacheprovider -k "window or overflow or soft_mixup or sharded" > $O/ft.log 2>&1; echo "ft rc=$?"; grep -n "fine-tune window\|grad-gate\|passed\|failed\|^E " $O/ft.log | cut -c1-400 | head -20
