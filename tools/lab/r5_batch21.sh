#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b21; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "big_tiles" -p no:cacheprovider 2>&1 | tail -3
timeout 900 python tools/lab/t_big_ab.py 2>&1 | grep "^M=" | tee $O/big_ab.txt
