#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b4; mkdir -p $O; cd $R
timeout 900 tools/lab/lab_big > $O/lab_big.log 2>&1; echo "lab_big rc=$?"; cat $O/lab_big.log
