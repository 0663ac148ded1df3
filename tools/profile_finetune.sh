#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ft; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o ft --output-format csv -- python3 $R/tools/bench_finetune.py --head CosFace --with-land ${1:-0} --dropout ${2:-0} --steps 10 --warmup 3 > $O/out.txt 2> $O/err.txt
tail -1 $O/out.txt
python3 - <<PY
import csv,glob,re
f=glob.glob("$O/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("kernel ms/step:", tot/13e6)
for r in rows[:28]:
    n=re.sub(r"\(anonymous namespace\)::|void ","",r["Name"])[:84]
    print(f"{n:84s} {int(r['Calls']):6d} {int(r['TotalDurationNs'])/13e6:8.3f} ms/step {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}%")
PY
