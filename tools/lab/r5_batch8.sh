#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b8; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -p no:cacheprovider -k "fused_head or big_tiles or dino_loss" > $O/kern.log 2>&1; echo "kern rc=$?"; tail -4 $O/kern.log | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_composition.py -x -q -p no:cacheprovider > $O/step.log 2>&1; echo "step rc=$?"; tail -4 $O/step.log | cut -c1-400
for v in 0 1 0 1; do
  LAFS_FUSED_HEAD=$v python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('FUSED_HEAD', $v, d['ms_per_step'], d['final_loss'])"
done
