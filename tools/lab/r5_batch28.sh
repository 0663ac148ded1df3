#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b28; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -q -s -x -p no:cacheprovider > $O/full.log 2>&1; echo "full rc=$?"; tail -4 $O/full.log | cut -c1-300
grep -h "grad-gate\|\[F17\]\|\[F18\]\|\[composition\]\|\[fine-tune window\]\|\[finetune graph\|\[row-chains\]\|\[eval-mode" $O/full.log | cut -c1-400 > $O/gates.txt; wc -l $O/gates.txt
LAFS_TEST_SHUFFLE=11 timeout 2400 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > $O/shuffle11.log 2>&1; echo "shuffle rc=$?"; tail -3 $O/shuffle11.log | cut -c1-300
python __graft_entry__.py smoke 2>&1 | tail -1
