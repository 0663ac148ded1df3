#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b26; mkdir -p $O; cd $R
timeout 900 python tools/lab/t_big_ab.py 2>&1 | grep "^M=\|Error\|error" | grep "GELU\|rror" | tee $O/big_ab.txt
