#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b6; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -p no:cacheprovider > $O/kern.log 2>&1; echo "kern rc=$?"; tail -4 $O/kern.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_finetune.py -x -q -s -p no:cacheprovider -k "window or overflow or soft_mixup or sharded" > $O/ft.log 2>&1; echo "ft rc=$?"; grep -n "fine-tune window\|grad-gate\|passed\|failed\|^E " $O/ft.log | cut -c1-300 | head -20
for v in 0 1 0 1; do
  LAFS_NT_BIG=$v timeout 600 python bench.py --extras-only mynet,finetune --no-roofline > $O/extras_big$v.json 2>/dev/null
  python - $O/extras_big$v.json $v <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])['extras']
print("NT_BIG", sys.argv[2], {k: v.get('ms_per_step', v) for k, v in d.items()})
PY
done
