#!/bin/bash
# GPU box: the order-dependent crash of the test suite: test_gpu_step.py followed by test_gpu_composition.py in ONE
# process, under different stream settings.  Prints the exit code of every variant.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/crash; mkdir -p $O; cd $R
run() { name=$1; shift; env "$@" timeout 900 python -X faulthandler -m pytest tests/test_gpu_step.py tests/test_gpu_composition.py -x -q -p no:cacheprovider > $O/$name.log 2>&1; echo "$name rc=$?"; tail -3 $O/$name.log | cut -c1-300; }
run default LAFS_DUMMY=1
run single_stream LAFS_SINGLE_STREAM=1
run no_chains LAFS_ROW_CHAINS=0 LAFS_ATTN_STREAM=0
