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
R = sys.argv[1] if len(sys.argv) > 1 else "round3"
DOM = "gemm_kres_kernel<1, true"          # student fc1: BF16_GELU epilogue (pre-activation stored) on the K-resident kernel
MLP = "mlp_fused_kernel<1, true"          # student saving forward with the LayerNorm prologue (csrc/mlp_fused.hip)
FC2 = "gemm_nt_kernel<2, 2, 64"           # student fc2: RESID_F32 epilogue, 128x128 tile, 64-deep stages (prefix: further template arguments follow)


def pmc_per_launch(path, name_parts, counters, launches_per_group=1):
    """Sum of each counter over the dispatches whose kernel name contains any of name_parts, divided by the number of GROUPS
    (one group = launches_per_group dispatches of the first name, e.g. a weight-gradient launch + its fold kernel)."""
    acc, n_first = {c: 0.0 for c in counters}, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] not in acc or not any(p in r["Kernel_Name"] for p in name_parts):
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        if name_parts[0] in r["Kernel_Name"] and r["Counter_Name"] == counters[0]:
            n_first += 1
    groups = max(n_first // launches_per_group, 1)
    return {c: acc[c] / groups for c in counters}, groups


SQN = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
       "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]


def kernel_entry(title, parts, alg_bytes, flops):
    f, n = pmc_per_launch(os.path.join(SRC, "pmc_fetch", "f_counter_collection.csv"), parts, ["FETCH_SIZE"])
    w, _ = pmc_per_launch(os.path.join(SRC, "pmc_write", "w_counter_collection.csv"), parts, ["WRITE_SIZE"])
    sq, _ = pmc_per_launch(os.path.join(SRC, "pmc_sq", "s_counter_collection.csv"), parts[:1], SQN)
    rd, wr = f["FETCH_SIZE"] * 1024 * 2, w["WRITE_SIZE"] * 1024
    wc = max(sq["SQ_WAVE_CYCLES"], 1)
    return {"kernel": title, "launches_averaged": n, "FETCH_SIZE_KB_raw": f["FETCH_SIZE"], "WRITE_SIZE_KB_raw": w["WRITE_SIZE"],
            "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
            "algorithmic_bytes_per_launch": alg_bytes, "flops_per_launch": flops, "sq_first_kernel": sq,
            "derived": {"wave_wait_frac": sq["SQ_WAIT_ANY"] / wc, "wave_issue_stall_frac": sq["SQ_WAIT_INST_ANY"] / wc,
                        "wave_active_frac": sq["SQ_ACTIVE_INST_ANY"] / wc, "lds_bank_conflict_cycles": sq["SQ_LDS_BANK_CONFLICT"]}}


def big_entry(M, geo):
    """The Part-fViT fc1 input gradient (M x 768 x 2048) on the persistent-tile kernel: the gemm_big_kernel dispatches whose name carries
    this geometry (two row counts run in one profile: they differ by their Geo<...>)."""
    def per_launch(path, counter):
        tot, n = 0.0, 0
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and "gemm_big_kernel" in r["Kernel_Name"] and geo in r["Kernel_Name"].replace(" ", ""):
                tot += float(r["Counter_Value"]); n += 1
        return (tot / n if n else None), n
    f, n = per_launch(os.path.join(SRC, "pmc_fetch", "f_counter_collection.csv"), "FETCH_SIZE")
    w, _ = per_launch(os.path.join(SRC, "pmc_write", "w_counter_collection.csv"), "WRITE_SIZE")
    if f is None or w is None:
        return None
    rd, wr = f * 1024 * 2, w * 1024
    return {"kernel": f"gemm_big_kernel<BF16, {geo}>  M={M} N=768 K=2048 (Part-fViT fc1 input gradient)", "launches_averaged": n,
            "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
            "algorithmic_bytes_per_launch": (M * 2048 + 768 * 2048 + M * 768) * 2.0, "flops_per_launch": 2.0 * M * 768 * 2048}


def main():
    os.makedirs(DST, exist_ok=True)
    shutil.copy(os.path.join(SRC, "step", "step_kernel_stats.csv"), os.path.join(DST, f"{R}_step_kernel_stats.csv"))
    shutil.copy(os.path.join(SRC, "step_bench.json"), os.path.join(DST, f"{R}_step_bench_under_rocprof.json"))
    shutil.copy(os.path.join(SRC, "dom", "dom_kernel_stats.csv"), os.path.join(DST, f"{R}_roofline_kernels_stats.csv"))
    shutil.copy(os.path.join(SRC, "dom_bench.json"), os.path.join(DST, f"{R}_roofline_kernels_bench.json"))
    M, D, I, H = 44160, 384, 384, 1536
    pairs = [(D, H), (H, D), (D, I), (3 * I, D)]
    sha_file = os.path.join(SRC, "csrc_sha.txt")
    out = {
        "csrc_sha": open(sha_file).read().strip() if os.path.isfile(sha_file) else None,
        "command": "rocprofv3 --kernel-trace --pmc <counter group> -- python3 bench.py --roofline-only   (one pass per group: FETCH_SIZE | "
                   "WRITE_SIZE | SQ_*)",
        "gfx950_correction": "FETCH_SIZE counts 128-B requests at 64 B for wide coalesced streams -> x2 (MI355X_MICROARCH.md, HBM "
                             "section); WRITE_SIZE used as reported",
        "wgrad_group": kernel_entry("wgrad_kernel<2,2,3,3,5> + wgrad_fold_kernel: four weight gradients of one ViT-S block, M=44160",
                                    ["wgrad_kernel", "wgrad_fold_kernel"],
                                    sum(2.0 * M * (a + b) + 4.0 * a * b for a, b in pairs), sum(2.0 * M * a * b for a, b in pairs)),
        "fc1": kernel_entry("gemm_kres_kernel<BF16_GELU>  M=44160 N=1536 K=384 (student fc1 forward, K-resident kernel)", [DOM],
                            (M * 384 + 1536 * 384 + 2 * M * 1536) * 2.0, 2.0 * M * 1536 * 384),
        "fc2": kernel_entry("gemm_nt_kernel<RESID_F32, WM=2, BK=64>  M=44160 N=384 K=1536 (student fc2 forward, tiled kernel)", [FC2],
                            (M * 1536 + 384 * 1536) * 2.0 + 2.0 * M * 384 * 4.0, 2.0 * M * 384 * 1536),
        "mlp_fused": kernel_entry("mlp_fused_kernel<FWD_SAVE, LN prologue>  M=25216 H=1536 (student LayerNorm 2 + MLP + residual + next LayerNorm 1, one launch)", [MLP],
                                  25216 * 384 * 4.0 * 2 + 2 * 1536 * 384 * 2.0 + 2 * 25216 * 1536 * 2.0 + 2 * (25216 * 384 * 2.0 + 25216 * 8.0),
                                  4.0 * 25216 * 384 * 1536),
    }
    for M, geo in ((44160, "Geo<11,4,1,4>"), (25216, "Geo<5,8,2,2>")):       # the extras' roofline kernel (bench.py roofline_partfvit_dgrad)
        try:
            e = big_entry(M, geo)
            if e is not None:
                out[f"partfvit_dgrad_{M}"] = e
        except Exception as ex:
            print("big_entry", M, ex)
    json.dump(out, open(os.path.join(DST, f"{R}_kernel_pmc.json"), "w"), indent=1)
    for name, sub in (("serial", "serial"), ("landmark_cnn", "cnn")):
        hits = glob.glob(os.path.join(ROOT, "gpurun_out", sub, "**", "*kernel_stats.csv"), recursive=True)
        if hits:
            shutil.copy(hits[0], os.path.join(DST, f"{R}_{name}_kernel_stats.csv"))
    hb = os.path.join(ROOT, "gpurun_out", "stephbm", "step_hbm.json")
    if os.path.isfile(hb):
        d = json.load(open(hb))
        d["hbm_bytes_per_step"] = int(d["hbm_GB_per_step"] * 1e9)
        json.dump(d, open(os.path.join(DST, f"{R}_step_hbm_pmc.json"), "w"), indent=1)
    for log, name in (("gemm.log", "gemm_pmc.txt"), ("attn.log", "attention_pmc.txt"), ("serial.log", "serial_kernel_table.txt")):
        src = os.path.join(ROOT, "gpurun_out", log)
        if os.path.isfile(src):
            shutil.copy(src, os.path.join(DST, f"{R}_{name}"))
    # serialised (one stream, no hipGraph) kernel tables of the extra workloads: tools/profile_extras_r5.sh
    for tab in glob.glob(os.path.join(ROOT, "gpurun_out", "r5", "*_serial_kernel_table.txt")):
        shutil.copy(tab, os.path.join(DST, f"{R}_{os.path.basename(tab)}"))
    # the kernels of ONE graph-replayed step (between two optimizer launches of the trace), with counts: shows what is -- and is
    # not (at::native, Tensile) -- inside a step, which the whole-run stats cannot (model construction launches ATen kernels)
    tr = glob.glob(os.path.join(SRC, "step", "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        import collections, re
        rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))
        marks = [i for i, r in enumerate(rows) if "zero_chunks" in r["Kernel_Name"]]      # first kernel of a step
        step = rows[marks[-2]:marks[-1]]
        cnt = collections.Counter(re.sub(r"\(anonymous namespace\)::|void |\(.*", "", r["Kernel_Name"]) for r in step)
        span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
        with open(os.path.join(DST, f"{R}_one_step_kernels.txt"), "w") as fo:
            fo.write(f"# {len(step)} kernel launches in one hipGraph-replayed step ({span:.2f} ms from first start to last end, under "
                     f"rocprofv3); foreign (at::native / Cijk / rocclr) kernels: "
                     f"{sorted(k for k in cnt if 'at::' in k or 'Cijk' in k or 'rocclr' in k)}\n")
            for k, v in cnt.most_common():
                fo.write(f"{v:5d}  {k}\n")
    print(json.dumps({k: {kk: out[k][kk] for kk in ("launches_averaged", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch")}
                      for k in ("wgrad_group", "fc1", "fc2", "mlp_fused")}))


if __name__ == "__main__":
    main()
