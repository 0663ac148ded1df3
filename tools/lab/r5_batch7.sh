#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b7; mkdir -p $O; cd $R
for rep in 1 2; do for v in lab_kres_old lab_kres lab_kres_noslp; do
  LAB_PHASES_ONLY=1 timeout 300 tools/lab/$v 2>&1 | grep -E "^(qkv|fc1|dgelu|proj) +abl0 " | sed "s/^/$v /"
done; done
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -p no:cacheprovider -k "k_resident or epilogues or big_tiles" > $O/kern.log 2>&1; echo "kern rc=$?"; tail -3 $O/kern.log | cut -c1-300
