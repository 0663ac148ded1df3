#!/bin/bash
# GPU box: HBM bytes moved by ONE training step (PMC FETCH_SIZE / WRITE_SIZE summed over every kernel of a step; eager
# launches so that each kernel is its own dispatch record; separate passes per counter as the TCC slots require).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stephbm; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; export LAFS_SINGLE_STREAM=1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-roofline --no-graph > $O/f.json 2> $O/f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-roofline --no-graph > $O/w.json 2> $O/w.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES -d $O/m -o m --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-roofline --no-graph > $O/m.json 2> $O/m.err
python3 - <<PY
import csv, glob, json
out = {}
# MFMA pipe occupancy of one step: SQ_VALU_MFMA_BUSY_CYCLES counts 16 cycles per v_mfma_f32_16x16x32_bf16 on its SIMD
f = glob.glob("$O/m/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
marks = [i for i, r in enumerate(rows) if "zero_chunks" in r["Kernel_Name"]]        # first kernel of a step
step = rows[marks[-2]:marks[-1]]
busy = sum(float(r["Counter_Value"]) for r in step)
dur_ns = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
out["mfma_busy_cycles_per_step"] = busy
out["kernel_time_ms_per_step_serialised"] = dur_ns / 1e6
out["mfma_pipe_busy_frac"] = busy / (dur_ns * 1e-9 * 2.4e9 * 1024)      # 256 CUs x 4 SIMDs at 2.4 GHz
out["mfma_flops_per_step_from_counter"] = busy / 16.0 * 16384
for tag, ctr in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    f = glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "zero_chunks" in r["Kernel_Name"]]
    a, b = marks[-2], marks[-1]                       # one full step: from its first kernel (gradient zeroing) to the next step's
    step = rows[a:b]
    out[ctr + "_KB_per_step"] = sum(float(r["Counter_Value"]) for r in step)
    out[ctr + "_kernels_per_step"] = len(step)
    by = {}
    for r in step:
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        by[k] = by.get(k, 0.0) + float(r["Counter_Value"])
    out[ctr + "_top"] = sorted(((round(v / 1e6, 3), k) for k, v in by.items()), reverse=True)[:12]
rd = out["FETCH_SIZE_KB_per_step"] * 1024 * 2         # gfx950: FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md)
wr = out["WRITE_SIZE_KB_per_step"] * 1024
out["hbm_read_GB_per_step"], out["hbm_write_GB_per_step"], out["hbm_GB_per_step"] = rd / 1e9, wr / 1e9, (rd + wr) / 1e9
import sys
sys.path.insert(0, "$R")
import bench
out["csrc_sha"] = bench.csrc_fingerprint()          # the kernel sources these bytes were measured on
json.dump(out, open("$O/step_hbm.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.endswith("_top")}))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    print(ctr, "top (GB raw KB/1e6):", out[ctr + "_top"][:8])
PY
