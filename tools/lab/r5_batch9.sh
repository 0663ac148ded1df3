#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b9; mkdir -p $O; cd $R
python tools/lab/t_head_loss.py 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/hl -o hl --output-format csv -- python3 $R/tools/lab/t_head_loss.py > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/hl/**/*kernel_stats.csv", recursive=True)[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
