#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/collect_profiles.sh / profile_serial.sh / profile_cnn.sh (gpurun_out/, scratch) into
the committed summaries under profiles/ (run in the build container after a gpurun call).  usage: tools/publish_profiles.py [round]"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "profiles"), os.path.join(ROOT, "profiles")
R = sys.argv[1] if len(sys.argv) > 1 else "round1"
DOM = "gemm_nt_kernel<1, 4, 32"           # BF16_GELU epilogue, 256x128 tile, 32-deep stages (prefix: further template arguments follow)


def per_launch(path, counters):
    """average counter value per launch of the dominant kernel (rocprofv3 emits one row per dispatch and counter)."""
    acc, n = {c: 0.0 for c in counters}, {c: 0 for c in counters}
    for r in csv.DictReader(open(path)):
        if DOM in r["Kernel_Name"] and r["Counter_Name"] in acc:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    return {c: acc[c] / max(n[c], 1) for c in counters}, max(n.values())


def main():
    os.makedirs(DST, exist_ok=True)
    shutil.copy(os.path.join(SRC, "step", "step_kernel_stats.csv"), os.path.join(DST, f"{R}_step_kernel_stats.csv"))
    shutil.copy(os.path.join(SRC, "step_bench.json"), os.path.join(DST, f"{R}_step_bench_under_rocprof.json"))
    shutil.copy(os.path.join(SRC, "dom", "dom_kernel_stats.csv"), os.path.join(DST, f"{R}_dominant_kernel_stats.csv"))
    shutil.copy(os.path.join(SRC, "dom_bench.json"), os.path.join(DST, f"{R}_dominant_kernel_bench.json"))
    f, nl = per_launch(os.path.join(SRC, "pmc_fetch", "f_counter_collection.csv"), ["FETCH_SIZE"])
    w, _ = per_launch(os.path.join(SRC, "pmc_write", "w_counter_collection.csv"), ["WRITE_SIZE"])
    sqn = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
           "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
    sq, _ = per_launch(os.path.join(SRC, "pmc_sq", "s_counter_collection.csv"), sqn)
    M, N, K = 44160, 1536, 384
    rd, wr = f["FETCH_SIZE"] * 1024 * 2, w["WRITE_SIZE"] * 1024
    out = {
        "kernel": "gemm_nt_kernel<BF16_GELU, WM=4, BK=32>  M=44160 N=1536 K=384 (student fc1 forward)",
        "command": "rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py --roofline-only   (one --pmc pass per counter group)",
        "launches_averaged": nl,
        "FETCH_SIZE_KB_raw": f["FETCH_SIZE"], "WRITE_SIZE_KB_raw": w["WRITE_SIZE"],
        "gfx950_correction": "FETCH_SIZE counts 128-B requests at 64 B for wide coalesced streams -> x2 (MI355X_MICROARCH.md, HBM "
                             "section); WRITE_SIZE used as reported",
        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": (M * K + N * K + 2 * M * N) * 2,
        "sq": sq,
        "derived": {"wave_wait_frac": sq["SQ_WAIT_ANY"] / max(sq["SQ_WAVE_CYCLES"], 1),
                    "wave_issue_stall_frac": sq["SQ_WAIT_INST_ANY"] / max(sq["SQ_WAVE_CYCLES"], 1),
                    "wave_active_frac": sq["SQ_ACTIVE_INST_ANY"] / max(sq["SQ_WAVE_CYCLES"], 1),
                    "lds_bank_conflict_cycles": sq["SQ_LDS_BANK_CONFLICT"]},
    }
    json.dump(out, open(os.path.join(DST, f"{R}_dominant_kernel_pmc.json"), "w"), indent=1)
    for name, sub in (("serial", "serial"), ("landmark_cnn", "cnn")):
        hits = glob.glob(os.path.join(ROOT, "gpurun_out", sub, "**", "*kernel_stats.csv"), recursive=True)
        if hits:
            shutil.copy(hits[0], os.path.join(DST, f"{R}_{name}_kernel_stats.csv"))
    print(json.dumps({k: out[k] for k in ("launches_averaged", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch")}))


if __name__ == "__main__":
    main()
