#!/bin/bash
# usage (on the GPU box): tools/lab/pmc.sh <binary | script.py> [args]   -> per-kernel SQ counter table (two passes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/labpmc; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
BIN=$R/$1; shift
case "$BIN" in *.py) set -- "$BIN" "$@"; BIN=python3;; esac
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/p1 -o g --output-format csv -- $BIN "$@" > $O/out1.txt 2> $O/err1.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/p2 -o g --output-format csv -- $BIN "$@" > $O/out2.txt 2> $O/err2.txt
python3 - <<PY
import csv, glob, collections, re
for pas in ("p1", "p2"):
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % pas, recursive=True)
    if not fs:
        print("no counters for", pas); print(open("$O/err%s.txt" % pas[1]).read()[-2000:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        key = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", r["Kernel_Name"])
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[key] += 1; dur[key] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for key, c in acc.items():
        n = max(cnt[key], 1); wc = c["SQ_WAVE_CYCLES"] / n
        rest = "  ".join(f"{k[3:]} {v/n/wc:.3f}" if k not in ("SQ_WAVE_CYCLES",) else f"WAVE_CYC {v/n:.3g}" for k, v in sorted(c.items()))
        print(f"{key:60s} n {n:3d} {dur[key]/n/1e3:7.1f} us  {rest}")
PY
