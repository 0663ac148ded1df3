#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b11; mkdir -p $O; cd $R
LAFS_TEST_SHUFFLE=5 timeout 2400 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > $O/shuffle5.log 2>&1; echo "shuffle5 rc=$?"; tail -6 $O/shuffle5.log | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -x -q -s -p no:cacheprovider > $O/full.log 2>&1; echo "full rc=$?"; tail -4 $O/full.log | cut -c1-300
grep -h "grad-gate\|\[F17\]\|\[composition\]\|\[fine-tune window\]\|\[finetune graph" $O/full.log > $O/gates.txt; wc -l $O/gates.txt
