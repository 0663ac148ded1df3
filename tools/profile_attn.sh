#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attn; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/p -o a --output-format csv -- python3 $R/tools/bench_kernels.py attn > $O/out.txt 2> $O/err.txt
tail -3 $O/out.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/p/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "attn_" not in k: continue
    import re
    key = (re.sub(r"\(anonymous namespace\)::|void |\(.*", "", k), r["Grid_Size"])
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[key] += 1
for key, c in sorted(acc.items()):
    n = max(cnt[key], 1); wc = c["SQ_WAVE_CYCLES"] / n
    print(key, "launches", n, " wave_cycles/launch %.3g" % wc, " wait_any %.2f  wait_inst %.2f (lds %.2f)  active %.2f  mfma_busy_cyc %.3g  lds_conflict %.3g" % (
        c["SQ_WAIT_ANY"] / n / wc, c["SQ_WAIT_INST_ANY"] / n / wc, c["SQ_WAIT_INST_LDS"] / n / wc, c["SQ_ACTIVE_INST_ANY"] / n / wc,
        c["SQ_VALU_MFMA_BUSY_CYCLES"] / n, c["SQ_LDS_BANK_CONFLICT"] / n))
PY
