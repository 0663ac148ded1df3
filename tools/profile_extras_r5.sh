#!/bin/bash
# GPU box: per-kernel EXCLUSIVE time of one of bench.py's extra workloads (default: the C4 fine-tune step), issued eagerly on ONE stream
# (LAFS_SINGLE_STREAM=1, no hipGraph) under rocprofv3 --kernel-trace --stats; writes gpurun_out/r5/<workload>_serial_kernel_table.txt
# (copied to profiles/round5_<workload>_serial_kernel_table.txt by tools/publish_profiles.py).
# usage: tools/profile_extras_r5.sh [finetune|mynet|finetune_plain|partialfc]
W=${1:-finetune}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5/prof_$W; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export LAFS_SINGLE_STREAM=1 LAFS_BENCH_EXTRAS_NO_GRAPH=1
rocprofv3 --kernel-trace --stats -d $O -o ft --output-format csv -- python3 $R/bench.py --extras-only $W --no-roofline > $O/bench.json 2> $O/err.txt
python3 - <<PY > $R/gpurun_out/r5/${W}_serial_kernel_table.txt
import csv, glob, re, json
print("# LAFS_SINGLE_STREAM=1, no hipGraph, rocprofv3 --kernel-trace --stats: exclusive kernel time per step")
print(open("$O/bench.json").read().strip()[:700])
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 16
tot = sum(int(r["TotalDurationNs"]) for r in rows)
aten = sum(int(r["TotalDurationNs"]) for r in rows if "at::native" in r["Name"])
print("total kernel ms per step (%d steps): %.2f   of which at::native kernels: %.3f ms/step (%d distinct)" % (steps, tot / steps / 1e6, aten / steps / 1e6, sum(1 for r in rows if "at::native" in r["Name"])))
fam = {}
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])
    k = re.sub(r"<.*", "", n.split("(")[0])
    fam[k] = fam.get(k, 0) + int(r["TotalDurationNs"])
print("by family: " + ", ".join("%s %.2f" % (k, v / steps / 1e6) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:10]))
for r in rows[:60]:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:100]
    print(f"{n:100s} {int(r['Calls']):6d} {int(r['TotalDurationNs']) / steps / 1e6:8.3f} ms/step {float(r['AverageNs']) / 1e3:9.1f} us")
PY
head -30 $R/gpurun_out/r5/${W}_serial_kernel_table.txt | cut -c1-170
