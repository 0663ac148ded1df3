#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gemm; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/p -o g --output-format csv -- python3 $R/tools/bench_kernels.py nt wg wgg > $O/out.txt 2> $O/err.txt
python3 - <<PY
import csv, glob, collections, re
f = glob.glob("$O/p/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int); dur = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "gemm_" not in k and "wgrad_" not in k: continue
    key = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", k) + " grid " + r["Grid_Size"]
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        cnt[key] += 1; dur[key] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for key, c in sorted(acc.items(), key=lambda kv: -dur[kv[0]]):
    n = max(cnt[key], 1); wc = c["SQ_WAVE_CYCLES"] / n
    mf = c["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (dur[key] / n * 1e-9 * 2.4e9 * 1024)
    print(f"{key:52s} n {n:3d} {dur[key]/n/1e3:7.1f} us  wait_any {c['SQ_WAIT_ANY']/n/wc:.2f} wait_inst {c['SQ_WAIT_INST_ANY']/n/wc:.2f} (lds {c['SQ_WAIT_INST_LDS']/n/wc:.2f}) active {c['SQ_ACTIVE_INST_ANY']/n/wc:.2f} mfma_busy {mf:.3f} ldsconf {c['SQ_LDS_BANK_CONFLICT']/n:.0f}")
PY
