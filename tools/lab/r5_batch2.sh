#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b2; mkdir -p $O; cd $R
for v in "" noslp "" noslp; do
  LAFS_LIB_VARIANT=$v python bench.py --no-extras --no-cpu-baseline --steps 40 > $O/bench_${v:-base}_$RANDOM.json 2> /dev/null
done
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r=d['roofline']
print(sys.argv[1].split('/')[-1], d['ms_per_step'], 'wgrad', r['avg_launch_us'], [(o['kernel'][:16], o['avg_launch_us']) for o in r['others']])
PY
done
LAFS_LIB_VARIANT=noslp timeout 600 python bench.py --extras-only mynet,finetune --no-roofline > $O/extras_noslp.json 2>/dev/null; tail -c 600 $O/extras_noslp.json; echo
timeout 600 python bench.py --extras-only mynet,finetune --no-roofline > $O/extras_base.json 2>/dev/null; tail -c 600 $O/extras_base.json; echo
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"; tail -8 $O/tests.log | cut -c1-300
timeout 900 python -X faulthandler -m pytest tests/test_gpu_step.py tests/test_gpu_composition.py -x -q -p no:cacheprovider > $O/order.log 2>&1; echo "order rc=$?"; tail -3 $O/order.log | cut -c1-300
