#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cnn; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o cnn --output-format csv -- python3 $R/tools/bench_cnn.py --hip-only > $O/out.txt 2> $O/err.txt
cat $O/out.txt
python3 - <<PY
import csv,glob,re
f=glob.glob("$O/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]:
    n=re.sub(r"\(anonymous namespace\)::|void ","",r["Name"])[:80]
    print(f"{n:80s} {int(r['Calls']):6d} {int(r['TotalDurationNs'])/22e6:8.3f} ms/fwd {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}%")
PY
