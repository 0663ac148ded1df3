#!/bin/bash
# GPU box: kernel trace of the graph-replayed step (phase spans, tools/step_timeline.py) + fabric-read bytes of the grouped weight
# gradient with and without the XCD map (bench.py --roofline-only under --pmc FETCH_SIZE, one pass per setting)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3prof; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/step -o step --output-format csv -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > $O/step_bench.json 2> $O/step.err
python3 $R/tools/step_timeline.py $(find $O/step -name "*kernel_trace.csv" | head -1) > $O/timeline.txt 2>&1
cat $O/timeline.txt
for m in 0 1; do
  export LAFS_WGRAD_XCD_MAP=$m
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f$m -o f --output-format csv -- python3 $R/bench.py --roofline-only > /dev/null 2> $O/f$m.err
  python3 - <<PY
import csv, glob
f = glob.glob("$O/f$m/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "wgrad_kernel" in r["Kernel_Name"]]
v = [float(r["Counter_Value"]) for r in rows]
print("LAFS_WGRAD_XCD_MAP=$m: wgrad_kernel launches", len(v), "FETCH_SIZE x2 per launch = %.1f MB" % (sum(v) / len(v) * 1024 * 2 / 1e6))
PY
done
