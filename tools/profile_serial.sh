#!/bin/bash
# GPU box: per-kernel durations of the step with every stream serialised (no time-sharing between concurrent kernels).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/serial; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; export LAFS_SINGLE_STREAM=1
rocprofv3 --kernel-trace --stats -d $O -o serial --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline --no-graph > $O/bench.json 2> $O/err.txt
cat $O/bench.json | cut -c1-400
python3 - <<PY
import csv,glob,re
f=glob.glob("$O/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (13 steps):", tot/13e6)
for r in rows[:45]:
    n=re.sub(r"\(anonymous namespace\)::|void ","",r["Name"])[:90]
    print(f"{n:90s} {int(r['Calls']):6d} {int(r['TotalDurationNs'])/13e6:8.3f} ms/step {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}%")
PY
