#!/bin/bash
# GPU box: per-kernel time of one of bench.py's extra workloads (default: the C4 fine-tune step) under rocprofv3 --kernel-trace --stats;
# writes gpurun_out/r4/<workload>_kernel_table.txt (copied to profiles/round4_<workload>_kernel_table.txt by hand)
# usage: tools/profile_finetune_r4.sh [finetune|mynet|finetune_plain|partialfc]
W=${1:-finetune}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4/prof_$W; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o ft --output-format csv -- python3 $R/bench.py --extras-only $W --no-roofline > $O/bench.json 2> $O/err.txt
python3 - <<PY > $R/gpurun_out/r4/${W}_kernel_table.txt
import csv, glob, re, json
print(open("$O/bench.json").read().strip()[:600])
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 16
tot = sum(int(r["TotalDurationNs"]) for r in rows)
aten = sum(int(r["TotalDurationNs"]) for r in rows if "at::native" in r["Name"])
print("total kernel ms per step (%d steps): %.2f   of which at::native kernels: %.3f ms/step (%d distinct)" % (steps, tot / steps / 1e6, aten / steps / 1e6, sum(1 for r in rows if "at::native" in r["Name"])))
for r in rows[:60]:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:100]
    print(f"{n:100s} {int(r['Calls']):6d} {int(r['TotalDurationNs']) / steps / 1e6:8.3f} ms/step {float(r['AverageNs']) / 1e3:9.1f} us")
PY
head -50 $R/gpurun_out/r4/${W}_kernel_table.txt
