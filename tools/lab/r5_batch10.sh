#!/bin/bash
# same-box A/B: the round-4 tree (tools/lab/_r4tree, commit 6bc2d31) against the current tree, interleaved
R=$GRAFT_REPO_ROOT; cd $R
run() { (cd $1 && python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$2', d['ms_per_step'], d['final_loss'])"); }
for rep in 1 2 3; do run tools/lab/_r4tree r4; run . now; LAFS_FUSED_HEAD=0 run . now_unfused; done
