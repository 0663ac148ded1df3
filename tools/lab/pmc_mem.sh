#!/bin/bash
# usage (on the GPU box): tools/lab/pmc_mem.sh <binary> [args]   -> per-kernel HBM/L2 traffic (FETCH_SIZE x2-corrected, WRITE_SIZE, TCC hit rate)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/labmem; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
BIN=$R/$1; shift
case "$BIN" in *.py) set -- "$BIN" "$@"; BIN=python3;; esac
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p1 -o g --output-format csv -- $BIN "$@" > $O/out1.txt 2> $O/err1.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p2 -o g --output-format csv -- $BIN "$@" > $O/out2.txt 2> $O/err2.txt
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/p3 -o g --output-format csv -- $BIN "$@" > $O/out3.txt 2> $O/err3.txt
python3 - <<PY
import csv, glob, collections, re
tab = collections.defaultdict(dict)
for pas in ("p1", "p2", "p3"):
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % pas, recursive=True)
    if not fs:
        print("no counters for", pas); print(open("$O/err%s.txt" % pas[1]).read()[-1500:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int)); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        key = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", r["Kernel_Name"])
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key][r["Counter_Name"]] += 1
        dur[key] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for key, c in acc.items():
        for k, v in c.items():
            tab[key][k] = v / cnt[key][k]
        tab[key]["us_" + pas] = dur[key] / sum(cnt[key].values()) / 1e3
for key, c in tab.items():
    f = c.get("FETCH_SIZE", 0) * 1024 * 2 / 1e6; w = c.get("WRITE_SIZE", 0) * 1024 / 1e6
    hit = c.get("TCC_HIT_sum", 0); mis = c.get("TCC_MISS_sum", 0)
    print(f"{key:44s} {c.get('us_p1', 0):7.1f} us  HBM read {f:8.1f} MB (x2 corrected)  write {w:7.1f} MB  L2 hit {hit/(hit+mis+1e-9):.3f}  L2 req {c.get('TCC_REQ_sum',0)/1e6:.2f} M")
PY
