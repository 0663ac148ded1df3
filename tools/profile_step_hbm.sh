#!/bin/bash
# GPU box: HBM bytes moved by ONE training step (PMC FETCH_SIZE / WRITE_SIZE summed over every kernel of a step; eager
# launches so that each kernel is its own dispatch record; separate passes per counter as the TCC slots require).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stephbm; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; export LAFS_SINGLE_STREAM=1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/f.json 2> $O/f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/w.json 2> $O/w.err
python3 - <<PY
import csv, glob, json
out = {}
for tag, ctr in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    f = glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "clip_adamw_ema" in r["Kernel_Name"]]
    a, b = marks[-2], marks[-1]                       # one full step: after the previous optimizer kernel .. this one
    step = rows[a + 1:b + 1]
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
json.dump(out, open("$O/step_hbm.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.endswith("_top")}))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    print(ctr, "top (GB raw KB/1e6):", out[ctr + "_top"][:8])
PY
