#!/bin/bash
# GPU box: per-kernel time of one of bench.py's extra workloads (default: the C4 fine-tune step) under rocprofv3 --kernel-trace --stats
# usage: tools/profile_finetune_r3.sh [finetune|mynet|finetune_plain|partialfc]
W=${1:-finetune}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ft_$W; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o ft --output-format csv -- python3 $R/bench.py --extras-only $W > $O/bench.json 2> $O/err.txt
cat $O/bench.json | cut -c1-300
python3 - <<PY
import csv, glob, re
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 16
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (%d steps): %.2f" % (steps, tot / steps / 1e6))
for r in rows[:40]:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:100]
    print(f"{n:100s} {int(r['Calls']):6d} {int(r['TotalDurationNs']) / steps / 1e6:8.3f} ms/step {float(r['AverageNs']) / 1e3:9.1f} us")
PY
