#!/usr/bin/env python3
"""LAFS pre-training step benchmark on MI355X (BASELINE.json metric: face-crops/sec/node).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config C2/C3): ViT-S/8 student+teacher, DINO head K=100000, 2 global 112x112 + 8 local 48x48 synthetic
crops, batch 64 per GPU, drop_path 0.1, epoch >= 1 (last layer unfrozen), bf16 MFMA GEMMs with fp32 accumulate /
residual stream / optimizer.  One "step" = teacher fwd + student fwd + DINO loss + backward + RCCL gradient/center
all-reduce + per-tensor clip + AdamW + teacher EMA + center EMA, inputs resident in HBM.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMG = {"vit_small": 113.3e9, "vit_tiny": None}


def step_flops(arch_dims, B, n_local, K):
    """Algorithmic FLOPs of one step per GPU: 2*MAC, backward = 2x forward for the student."""
    D, depth, heads = arch_dims
    mlp = 4 * D
    per_tok_layer = 2 * (D * 3 * D + D * D + 2 * D * mlp)
    def crop(n_tok):
        return depth * (per_tok_layer * n_tok + 4 * n_tok * n_tok * D) + 2 * 192 * D * (n_tok - 1)
    head = 2 * (D * 2048 + 2048 * 2048 + 2048 * 256 + 256 * K)
    g, l = crop(197), crop(37)
    student = 3 * (2 * g + n_local * l + (2 + n_local) * head)
    teacher = 2 * g + 2 * head
    return B * (student + teacher)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)           # >= 50 timed steps after >= 10 warm-up
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--arch", default="vit_small")
    ap.add_argument("--out-dim", type=int, default=100000)
    ap.add_argument("--local-crops", type=int, default=8)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--depth", type=int, default=0, help="override the trunk depth (tests of the multi-rank launch structure at toy size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true", help="run only the roofline kernels' loops (for rocprofv3)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the roofline kernels' loops (profiles of the step alone)")
    ap.add_argument("--cpu-batch", type=int, default=64, help="batch of the CPU oracle leg (default: the metric's own batch; ~12 s per step on 32 threads)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the non-headline workloads appended to the JSON line at 1 GPU (`extras`: the reference's real "
                         "pre-training pair --arch mynet, and the C4 fine-tune step)")
    ap.add_argument("--extras-only", default="", help="run only this extra workload (mynet | finetune | finetune_plain | partialfc) and print it")
    ap.add_argument("--augment", action="store_true",
                    help="with --frontend: also run the device-side DataAugmentation_LAFS (uint8 batch -> 20 views) every step")
    ap.add_argument("--frontend", action="store_true",
                    help="include the landmark front-end (frozen MobileNetV3 on 10*B views + theta + gathers) in every step, "
                         "software-pipelined on its own stream; default off = landmark crops resident in HBM (the headline metric)")
    return ap.parse_args()


def _time_on_stream(fn, iters):
    """Average duration of fn() with HIP events on the stream the kernels are launched on (torch's current stream).
    The warm-up is as long as the timed region: the shader clock needs tens of ms of load to settle (from idle it climbs, and a
    sustained MFMA kernel then sits at the 1400 W cap around 1.57 GHz -- tools/lab/wgrad_clocks.sh), so a 30-launch sample
    right after a pause reads up to 15 % slower than the steady rate (tools/lab/NOTES.md)."""
    for _ in range(max(3, iters)):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def _roof(kernel, dur, flops, alg_bytes, traffic=None):
    """Both roofs of one launch: HBM (algorithmic bytes / time vs 8 TB/s) and MFMA (flops / time vs 2.5 PFLOP/s dense bf16).
    `bound` is the roof the kernel's arithmetic intensity puts it under (machine balance 312 FLOP/B); `frac` is against it."""
    gbs, tf = alg_bytes / dur / 1e9, flops / dur / 1e12
    bound = "hbm" if flops / alg_bytes < 2.5e15 / 8e12 else "mfma"
    return {"bound": bound, "kernel": kernel,
            "achieved": round(gbs if bound == "hbm" else tf, 1), "peak": 8000.0 if bound == "hbm" else 2500.0,
            "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": round(gbs / 8000.0 if bound == "hbm" else tf / 2500.0, 4),
            "traffic": traffic, "avg_launch_us": round(dur * 1e6, 2), "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / 8000.0, 4),
            "mfma_tflops": round(tf, 1), "mfma_frac": round(tf / 2500.0, 4), "intensity_flop_per_byte": round(flops / alg_bytes, 1)}


def csrc_fingerprint():
    """sha256 (12 hex digits) over the kernel sources the library is built from: profiles/*_pmc.json record the fingerprint they
    were measured on (tools/publish_profiles.py, tools/profile_step_hbm.sh), and numbers read from a profile are only printed while
    it still matches -- a kernel edit without a profile refresh turns them into null instead of a stale figure."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "lafs_cvpr2024_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def _latest_profile(suffix):
    """(path, parsed json, file sha) of the newest committed profiles/round<N>_<suffix>, or (None, None, None)."""
    import glob
    import hashlib
    import re
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_" + suffix)),
                  key=lambda f: int(re.search(r"round(\d+)_", os.path.basename(f)).group(1)))
    if not hits:
        return None, None, None
    try:
        raw = open(hits[-1], "rb").read()
        return hits[-1], json.loads(raw), hashlib.sha256(raw).hexdigest()[:12]
    except Exception:
        return None, None, None


PROFILE_NOTES = {}


def _fresh(tag, path, data, sha):
    """True when the profile was measured on the kernels this run uses; records provenance for the JSON line either way."""
    ok = data is not None and data.get("csrc_sha") == csrc_fingerprint()
    PROFILE_NOTES[tag] = {"file": os.path.relpath(path, ROOT) if path else None, "sha256": sha,
                          "measured_on_csrc": None if data is None else data.get("csrc_sha"), "current_csrc": csrc_fingerprint(),
                          "used": bool(ok)}
    return ok


def _pmc_traffic(name):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/round<N>_kernel_pmc.json), or None when there is
    no such profile or it was measured on other kernel sources than the ones built now."""
    path, data, sha = _latest_profile("kernel_pmc.json")
    if not _fresh("kernel_pmc", path, data, sha):
        return None
    try:
        return round(data[name]["hbm_bytes_per_launch"])
    except Exception:
        return None


def roofline_wgrad_group(device, iters=200):
    """The four weight gradients of one ViT-S block (student: M = 64*(2*197+8*37) = 44160 tokens) as the engine launches
    them: ONE lafs_wgrad_group call = wgrad_kernel<2,2,3,3,5> (240 workgroups: 48 tiles x 5 token slices) + the fold kernel.
    Algorithmic bytes per launch: every operand read once (8 bf16 matrices) + the four fp32 gradients written once."""
    from lafs_cvpr2024_amd import ops
    M, D, I, H = 44160, 384, 384, 1536
    mk = lambda r, c: torch.randn(r, c, device=device).to(torch.bfloat16)
    pairs = [(mk(M, D), mk(M, H)), (mk(M, H), mk(M, D)), (mk(M, D), mk(M, I)), (mk(M, 3 * I), mk(M, D))]
    probs = [(x, y, torch.zeros(x.shape[1], y.shape[1], device=device), True, torch.zeros(x.shape[1], device=device)) for x, y in pairs]
    ws = ops.wgrad_group(probs)
    dur = _time_on_stream(lambda: ops.wgrad_group(probs, workspace=ws), iters)
    flops = sum(2.0 * M * x.shape[1] * y.shape[1] for x, y in pairs)
    alg = sum(2.0 * M * (x.shape[1] + y.shape[1]) + 4.0 * x.shape[1] * y.shape[1] for x, y in pairs)
    return _roof("wgrad_kernel<2,2,3,3,5> + wgrad_fold_kernel: the 4 weight gradients of one ViT-S block, M=44160 (lafs_wgrad_group)",
                 dur, flops, alg, _pmc_traffic("wgrad_group"))


def roofline_fc1(device, iters=200):
    """Heaviest single launch of the NT GEMM family: student MLP fc1 (M = 44160, N = 1536, K = 384, bias+GELU epilogue writing
    u and gelu(u)).  Algorithmic bytes: A and W (bf16) read once, u and GELU(u) (bf16) written once."""
    from lafs_cvpr2024_amd import _lib, ops
    M, N, K = 44160, 1536, 384
    A = torch.randn(M, K, device=device).to(torch.bfloat16)
    W = (torch.randn(N, K, device=device) * 0.02).to(torch.bfloat16)
    b = torch.zeros(N, device=device)
    u = torch.empty(M, N, device=device, dtype=torch.bfloat16)
    a = torch.empty(M, N, device=device, dtype=torch.bfloat16)
    dur = _time_on_stream(lambda: ops.gemm_nt(A, W, _lib.EPI_BF16_GELU, bias=b, out=u, out2=a), iters)
    return _roof("gemm_kres_kernel<BF16_GELU> M=44160 N=1536 K=384 (student MLP fc1 + GELU, K-resident streaming kernel)", dur,
                 2.0 * M * N * K, (M * K + N * K + 2 * M * N) * 2.0, _pmc_traffic("fc1"))


def roofline_fc2(device, iters=200):
    """Heaviest launch left on the tiled NT kernel: student MLP fc2 forward (M = 44160, N = 384, K = 1536, residual epilogue:
    x = x1 + DropPath(fc2(a) + b)).  Algorithmic bytes: A and W (bf16) read once, the fp32 residual read once, the fp32 output
    written once."""
    from lafs_cvpr2024_amd import _lib, ops
    M, N, K = 44160, 384, 1536
    A = torch.randn(M, K, device=device).to(torch.bfloat16)
    W = (torch.randn(N, K, device=device) * 0.02).to(torch.bfloat16)
    b = torch.zeros(N, device=device)
    x1 = torch.randn(M, N, device=device)
    out = torch.empty(M, N, device=device)
    dur = _time_on_stream(lambda: ops.gemm_nt(A, W, _lib.EPI_RESID_F32, bias=b, resid=x1, out=out), iters)
    return _roof("gemm_nt_kernel<RESID_F32,2,64> M=44160 N=384 K=1536 (student MLP fc2 + residual, tiled LDS-DMA kernel)", dur,
                 2.0 * M * N * K, (M * K + N * K) * 2.0 + 2.0 * M * N * 4.0, _pmc_traffic("fc2"))


def roofline_mlp_fused(device, iters=200):
    """Heaviest launch of the fused-MLP family (csrc/mlp_fused.hip): the student's saving forward on the long row chain (M = 128 x 197
    = 25 216 rows = 197 workgroups), LayerNorm 2 in the prologue and the next block's LayerNorm 1 in the epilogue, as the engine
    launches it in 11 of 12 blocks.  Algorithmic bytes: the fp32 residual stream read once (it is LayerNorm input and residual),
    both weight matrices once, the fp32 result, gelu'(u) and gelu(u) (bf16), the two bf16 LayerNorm outputs and their row
    statistics written once; the 1536-wide intermediate never leaves the chip."""
    from lafs_cvpr2024_amd import _lib, ops
    M, D, H = 25216, 384, 1536
    x1 = torch.randn(M, D, device=device)
    W1 = (torch.randn(H, D, device=device) * 0.02).to(torch.bfloat16)
    W2 = (torch.randn(D, H, device=device) * 0.02).to(torch.bfloat16)
    b1, b2 = torch.zeros(H, device=device), torch.zeros(D, device=device)
    gam, bet = torch.ones(D, device=device), torch.zeros(D, device=device)
    out = torch.empty(M, D, device=device)
    g = torch.empty(M, H, device=device, dtype=torch.bfloat16); a = torch.empty(M, H, device=device, dtype=torch.bfloat16)
    h = torch.empty(M, D, device=device, dtype=torch.bfloat16); st = torch.empty(M, 2, device=device)
    hn = torch.empty(M, D, device=device, dtype=torch.bfloat16); sn = torch.empty(M, 2, device=device)
    dur = _time_on_stream(lambda: ops.mlp_fused(None, W1, W2, _lib.MLP_FWD_SAVE, bias_a=b1, bias_b=b2, resid=x1, out=out, save_grad=g,
                                                save_act=a, ln=(gam, bet, 1e-6), ln_stats=st, ln_out=h,
                                                next_ln=(gam, bet, 1e-6, hn, sn)), iters)
    alg = M * D * 4.0 + 2 * H * D * 2.0 + M * D * 4.0 + 2 * M * H * 2.0 + 2 * (M * D * 2.0 + M * 8.0)
    return _roof("mlp_fused_kernel<FWD_SAVE, LN prologue> M=25216 H=1536 (student LayerNorm 2 + fc1 + GELU + fc2 + residual + next LayerNorm 1 in one launch)",
                 dur, 4.0 * M * D * H, alg, _pmc_traffic("mlp_fused"))


ROOFLINE_KERNELS = {"wgrad_kernel": roofline_wgrad_group, "gemm_nt_kernel": roofline_fc2, "gemm_kres_kernel": roofline_fc1,
                    "mlp_fused_kernel": roofline_mlp_fused}


def kernel_families():
    """{family: ms per step} from the newest committed SERIALISED profile of the step (profiles/round<N>_serial_kernel_stats.csv:
    rocprofv3 --kernel-trace --stats with LAFS_SINGLE_STREAM=1 --no-graph, exclusive times).  A family = every template instantiation
    of one kernel (the name up to '<'): the K = 384 GEMMs are five instantiations of gemm_kres_kernel and lead the step as a set
    although no single one of them does."""
    import csv
    import glob
    import re
    fam = {}
    f = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_serial_kernel_stats.csv")),
               key=lambda x: int(re.search(r"round(\d+)_", os.path.basename(x)).group(1)))[-1]
    rows = list(csv.DictReader(open(f)))
    # steps in the trace: the per-step zeroing kernel runs once per step
    steps = max([int(r["Calls"]) for r in rows if "zero_chunks_kernel" in r["Name"]] or [1])
    for r in rows:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        key = name.split("<")[0].split("(")[0].strip()
        fam[key] = fam.get(key, 0.0) + float(r["TotalDurationNs"]) / 1e6 / steps
    return fam, os.path.relpath(f, ROOT)


def dominant_kernel_name():
    """The kernel FAMILY with the largest per-step total in the newest committed serialised profile of the step (all instantiations
    of a template summed); its heaviest launch is the one timed live.  Falls back to the weight-gradient kernel."""
    try:
        fam, _ = kernel_families()
        for key, _ms in sorted(fam.items(), key=lambda kv: -kv[1]):
            if key in ROOFLINE_KERNELS:
                return key
    except Exception:
        pass
    return "wgrad_kernel"


def dominant_kernel_roofline(device, iters=200):
    dom = dominant_kernel_name()
    out = ROOFLINE_KERNELS[dom](device, iters)
    try:                                             # how the families rank in that profile (ms of exclusive kernel time per step)
        fam, src = kernel_families()
        out["family"] = dom
        out["family_ms_per_step"] = {k: round(v, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:6]}
        out["family_profile"] = src
    except Exception:
        pass
    out["others"] = [fn(device, iters) for k, fn in ROOFLINE_KERNELS.items() if k != dom]
    return out


def step_hbm_bytes():
    """HBM bytes of ONE whole step from the newest profiles/round<N>_step_hbm_pmc.json (FETCH_SIZE x2 + WRITE_SIZE over all
    kernels of a step), or None when that profile does not describe the kernels built now."""
    path, data, sha = _latest_profile("step_hbm_pmc.json")
    if not _fresh("step_hbm_pmc", path, data, sha):
        return None
    try:
        return int(data["hbm_bytes_per_step"])
    except Exception:
        return None


def cpu_baseline(arch_dims, n_local, K, batch, timed=2):
    """The oracle (fp32 torch-CPU restatement, parity-locked to the reference's golden vectors) timed on the host cores on a
    bounded sample of the same workload: steps of the same model / crop geometry at batch `batch` -- ONE untimed warm-up step
    (thread pool start, allocator growth, first-touch of the 100 000-class head) and then `timed` timed steps, median reported."""
    from oracle import step as ostep, vit as ovit
    D, depth, heads = arch_dims
    # torch's CPU GEMMs stop scaling (and collapse when oversubscribed) beyond a few dozen threads at these sizes:
    # measured on the GPU box host, 32 threads is the fastest setting (256 threads is 100x slower)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)

    def run(cfg, b, nl, n_timed):
        st = ostep.LafsState(cfg, out_dim=K, seed=0)
        g = torch.Generator().manual_seed(0)
        crops = [torch.randn(b, 3, 112, 112, generator=g).clamp(-1, 1) for _ in range(2)] + \
                [torch.randn(b, 3, 48, 48, generator=g).clamp(-1, 1) for _ in range(nl)]
        ts, r = [], None
        for i in range(1 + n_timed):
            t0 = time.time()
            r = ostep.lafs_step(st, crops, epoch=1, lr=5e-4, wd=0.04, momentum=0.996, teacher_temp=0.04)
            if i > 0:
                ts.append(time.time() - t0)
        ts.sort()
        med = ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])
        return med, ts, float(r["loss"])

    cfg = ovit.ViTConfig(patch_size=8, embed_dim=D, depth=depth, num_heads=heads, img_size=224)
    med, ts, loss = run(cfg, batch, n_local, timed)
    out = {"value": round(batch * (2 + n_local) / med, 3), "unit": "face-crops/s", "cores": cores, "kind": "port",
           "sample": f"same model/crops at batch {batch}, fp32 torch CPU: 1 warm-up + {timed} timed steps, median {med:.2f} s "
                     f"(all: {', '.join('%.2f' % t for t in ts)}), loss {loss:.4f}"}
    # config C1 of BASELINE.json (ViT-Tiny, 2 global + 2 local crops, batch 8): the always-run CPU case, for round-to-round comparison
    cfg1 = ovit.ViTConfig(patch_size=8, embed_dim=192, depth=12, num_heads=3, img_size=224)
    med1, ts1, _ = run(cfg1, 8, 2, 3)
    out["c1"] = {"value": round(8 * 4 / med1, 3), "unit": "face-crops/s",
                 "sample": f"C1: ViT-Tiny, 2g+2l, batch 8, out_dim {K}: 1 warm-up + 3 timed steps, median {med1:.2f} s"}
    return out


def partfvit_pass_flops(n_seq, n_tok_seq, dim=768, heads=11, mlp=2048, depth=12):
    """Forward FLOPs of one Part-fViT trunk pass over n_seq sequences of n_tok_seq tokens (GEMMs + attention; the patch embedding
    192 -> dim included): the algorithmic count, 2 m n k per GEMM."""
    T, inner = n_seq * n_tok_seq, heads * 64
    per_layer = 2.0 * T * dim * (3 * inner) + 2.0 * T * inner * dim + 2 * 2.0 * T * dim * mlp + 4.0 * n_seq * heads * n_tok_seq * n_tok_seq * 64
    return depth * per_layer + 2.0 * T * 192 * dim


def roofline_partfvit_dgrad(device, M, iters=100):
    """Dominant kernel family of both Part-fViT workloads in the serialised profiles (profiles/round*_finetune_kernel_table.txt,
    round*_mynet_kernel_table.txt): the plain input-gradient GEMMs; the largest instance is the fc1 input gradient dH2 = dU W1
    (M tokens, N = 768, K = 2048), on whichever kernel lafs_gemm_nt routes it to (named in the line).  Algorithmic bytes: dU and
    W1^T (bf16) read once, dH2 (bf16) written."""
    from lafs_cvpr2024_amd import _lib, ops
    N, K = 768, 2048
    A = torch.randn(M, K, device=device).to(torch.bfloat16)
    W = (torch.randn(N, K, device=device) * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, device=device, dtype=torch.bfloat16)
    dur = _time_on_stream(lambda: ops.gemm_nt(A, W, _lib.EPI_BF16, out=out), iters)
    route = ops.gemm_nt(A, W, _lib.EPI_BF16, out=out, route_only=True)
    kern = {5: "gemm_big_kernel<BF16> 192x256 tiles, one persistent workgroup per CU", 4: "gemm_nt_kernel<BF16,2,64,MB=5> 160x128 tiles",
            3: "gemm_nt_kernel<BF16,2,64,6> 128x384 tiles"}.get(route, "gemm_nt_kernel<BF16,2,64> 128x128 tiles")
    return _roof(f"{kern} M={M} N=768 K=2048 (Part-fViT fc1 input gradient)", dur, 2.0 * M * N * K, (M * K + N * K + M * N) * 2.0,
                 _pmc_traffic(f"partfvit_dgrad_{M}"))


EXTRA_ROOFLINE = True            # --no-roofline: the extras' kernel tables under rocprofv3 must not contain the roofline loops


def _extras_graph():
    """The extra workloads run as one hipGraph per step; LAFS_BENCH_EXTRAS_NO_GRAPH=1 (profiling only: tools/profile_extras_r5.sh, with
    LAFS_SINGLE_STREAM=1) issues them eagerly on one stream so that rocprofv3's per-kernel times are exclusive times."""
    return os.environ.get("LAFS_BENCH_EXTRAS_NO_GRAPH") != "1"


def _extra_evidence(out, flops, dt, device, M):
    out["step_tflops"] = round(flops / dt / 1e12, 1)
    out["step_mfma_frac"] = round(flops / dt / 2.5e15, 4)
    out["step_flops_algorithmic"] = round(flops / 1e12, 3)
    if not EXTRA_ROOFLINE:
        return out
    try:
        out["roofline"] = roofline_partfvit_dgrad(device, M)
    except Exception as e:
        out["roofline"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def extra_mynet_pretrain(device, B=64, nl=8, K=100000, steps=12, warmup=4):
    """NON-headline: the pair the reference actually pre-trains (lafs_train.py:300-335, --arch mynet): Part-fViT dim 768 / depth 12 /
    11 heads / mlp 2048 student AND teacher with dropout 0.1 + DropPath 0.1 live in both, [B, n, 192] landmark-patch tokens,
    2 global + 8 local crops, batch 64, K = 100 000, one captured hipGraph (element-dropout seed read from the device counter)."""
    import types
    from lafs_cvpr2024_amd.dino_loss import DINOLoss
    from lafs_cvpr2024_amd.engine import LafsPretrainEngine
    from lafs_cvpr2024_amd.lafs_train import build_backbones
    from lafs_cvpr2024_amd.utils import MultiCropWrapper
    from lafs_cvpr2024_amd import vision_transformer as vits
    torch.manual_seed(0)
    sb, tb, dim = build_backbones(types.SimpleNamespace(arch="mynet", mynet_dims="768,12,11,2048", mynet_dropout=0.1))
    student = MultiCropWrapper(sb, vits.DINOHead(dim, K, use_bn=False, norm_last_layer=True))
    teacher = MultiCropWrapper(tb, vits.DINOHead(dim, K, use_bn=False))
    teacher.load_state_dict(student.state_dict())
    crit = DINOLoss(K, 2 + nl, 0.07, 0.04, 30, 41)
    eng = LafsPretrainEngine(student, teacher, crit, B, n_local=nl, clip_grad=3.0, freeze_last_layer=1, use_graph=_extras_graph(), device=device)
    g = torch.Generator(device=device).manual_seed(0)
    eng.in_global_all.copy_(torch.randn(eng.in_global_all.shape, device=device, generator=g).clamp_(-1, 1))
    eng.in_local_all.copy_(torch.randn(eng.in_local_all.shape, device=device, generator=g).clamp_(-1, 1))
    one = lambda: eng.step(lr=2.5e-4, wd=0.04, momentum=0.996, teacher_temp=0.04, epoch=1)
    for _ in range(warmup):
        one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    out = {"workload": f"Part-fViT (768/12/11/2048) LAFS pair, dropout 0.1 + DropPath 0.1 live in student and teacher, 2g+{nl}l crops, "
                       f"batch {B}, out_dim {K}, hipGraph {bool(eng._graphs)}", "ms_per_step": round(dt * 1e3, 2),
           "face_crops_per_s": round(B * (2 + nl) / dt, 1), "steps": steps, "warmup": warmup, "loss": round(float(loss.item()), 4)}
    # algorithmic FLOPs: student forward + backward (3x) over 2 global (197 tokens) + nl local (37 tokens) views, teacher forward over
    # the 2 global ones, DINO heads (dim -> 2048 -> 2048 -> 256 -> K) on 10 B / 2 B rows
    f_s = partfvit_pass_flops(2 * B, 197) + partfvit_pass_flops(nl * B, 37)
    f_t = partfvit_pass_flops(2 * B, 197)
    head = lambda rows: 2.0 * rows * (dim * 2048 + 2048 * 2048 + 2048 * 256 + 256 * K)
    flops = 3 * (f_s + head((2 + nl) * B)) + f_t + head(2 * B)
    return _extra_evidence(out, flops, dt, device, B * (2 * 197 + nl * 37))


def extra_finetune(device, head="CosFace", with_land=True, dropout=0.1, B=128, C=205990, steps=12, warmup=4):
    """NON-headline: BASELINE.json configs[3] on ONE GPU as train_largescale.py builds it (:432, 542-561, 785-891): Part-fViT ViT-B +
    CosFace over 205 990 classes, trainable landmark branch, dropout 0.1, mixup alpha 0.2 / prob 0.1, batch 128, acc_step 1 here
    (every micro-step followed by AdamW: the upper bound of the per-step cost).  head = PartialFC: configs[4]'s sampled head."""
    from lafs_cvpr2024_amd.face_pre_pro.ViT_face import ViT_face_landmark_patch8
    from lafs_cvpr2024_amd.finetune_engine import FinetuneEngine
    torch.manual_seed(0)
    sharded = head == "PartialFC"
    m = ViT_face_landmark_patch8(loss_type="None" if sharded else "CosFace", GPU_ID=None, num_class=C, image_size=112, patch_size=8,
                                 dim=768, depth=12, heads=11, mlp_dim=2048, dropout=dropout, emb_dropout=dropout,
                                 with_land=with_land, drop_path_rate=0.1)
    sh = None
    if sharded:
        from lafs_cvpr2024_amd.partial_fc import PartialFC
        sh = PartialFC(768, C, B, sample_rate=0.1, device=device)
    eng = FinetuneEngine(m, B, acc_step=1, device=device, sharded_head=sh, use_graph=_extras_graph())
    m.train()
    x = torch.randint(0, 256, (B, 3, 112, 112), dtype=torch.uint8, device=device)
    y = torch.randint(0, C, (B,), device=device)
    for _ in range(warmup):
        eng.step(x, y, lr=1e-4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        loss = eng.step(x, y, lr=1e-4)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    out = {"workload": f"Part-fViT ViT-B + {head} fine-tune step, batch {B}, {C} classes, with_land={with_land}, dropout={dropout}"
                       + (", sample_rate 0.1" if sharded else "") + f", hipGraph {bool(eng._graphs)}", "ms_per_step": round(dt * 1e3, 2),
           "images_per_s": round(B / dt, 1), "steps": steps, "warmup": warmup, "loss": round(float(loss.item()), 4)}
    # algorithmic FLOPs: trunk forward + backward (3x) + the class-table GEMMs (cos, d emb, d W: 3 x 2 B C D; the sampled head scores
    # 10 % of the centres); the landmark CNN (~0.3 GFLOP per image forward) is left out of the count
    cls = C * (0.1 if sharded else 1.0)
    flops = 3 * partfvit_pass_flops(B, 197) + 3 * 2.0 * B * cls * 768
    return _extra_evidence(out, flops, dt, device, B * 197)


EXTRAS = {"mynet": lambda d: extra_mynet_pretrain(d),
          "finetune": lambda d: extra_finetune(d),
          "finetune_plain": lambda d: extra_finetune(d, with_land=False, dropout=0.0),
          "partialfc": lambda d: extra_finetune(d, head="PartialFC", with_land=False, B=256, C=200000)}


def run_extras(device, which=("mynet", "finetune")):
    out = {}
    for k in which:
        torch.cuda.empty_cache()
        try:
            out[k] = EXTRAS[k](device)
        except Exception as e:                       # an extra must never cost the headline line
            out[k] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script (one per GPU, RCCL over
    127.0.0.1) BEFORE this process touches the GPU, relay rank 0's JSON line, exit with the worst return code."""
    import socket
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count()                 # counting devices does not initialise the GPU
    if have < n and os.environ.get("LAFS_BENCH_SHARE_GPU") != "1":
        sys.exit(f"bench.py: --gpus {n} but only {have} GPU(s) are visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.exit(max(abs(rc) for rc in rcs))


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            spawn_ranks(args)                        # never returns
        world, rank, local = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world} set by the launcher")
    share = os.environ.get("LAFS_BENCH_SHARE_GPU") == "1"     # tests: every rank on GPU 0 over gloo (RCCL wants one device per rank)
    if share:
        local = 0
    from lafs_cvpr2024_amd import _lib
    dbg = int(_lib.lib().lafs_debug_get())
    if dbg != 0 or _lib.lib().lafs_ablation_build():
        sys.exit(f"bench.py: refusing to run with kernel debug flags set (lafs_debug_get() = {dbg}, ablation build = "
                 f"{_lib.lib().lafs_ablation_build()}): unset LAFS_DEBUG_FLAGS / LAFS_USE_ABLATE_LIB")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from lafs_cvpr2024_amd import vision_transformer as vits
    from lafs_cvpr2024_amd.dino_loss import DINOLoss
    from lafs_cvpr2024_amd.engine import LafsPretrainEngine
    from lafs_cvpr2024_amd.utils import MultiCropWrapper, cosine_scheduler

    if args.roofline_only:
        # (the extras' dominant kernel at both workloads' row counts rides along, so that the PMC passes of tools/collect_profiles_round.sh
        # cover it: extras.*.roofline.traffic)
        print(json.dumps({"roofline": dominant_kernel_roofline(device, iters=200),
                          "extras_roofline": {str(M): roofline_partfvit_dgrad(device, M) for M in (44160, 25216)}}))
        return
    if args.no_roofline:
        global EXTRA_ROOFLINE
        EXTRA_ROOFLINE = False
    if args.extras_only:
        print(json.dumps({"extras": run_extras(device, [k for k in args.extras_only.split(",") if k])}))
        return
    # the number of ranks the communication backend really sees must be the number of GPUs the line will claim
    comm_ranks = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if comm_ranks != args.gpus:
        sys.exit(f"bench.py: the process group has {comm_ranks} rank(s) but --gpus is {args.gpus}")

    dims = {"vit_small": (384, 12, 6), "vit_tiny": (192, 12, 3), "vit_base": (768, 12, 12)}[args.arch]
    torch.manual_seed(0)
    B, K, nl = args.batch, args.out_dim, args.local_crops
    if args.depth > 0:                               # (tests only: the named architecture cut to --depth blocks)
        from functools import partial
        dims = (dims[0], args.depth, dims[2])
        mk = lambda dpr: vits.VisionTransformer(patch_size=8, embed_dim=dims[0], depth=dims[1], num_heads=dims[2], mlp_ratio=4, qkv_bias=True,
                                                norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), drop_path_rate=dpr)
    else:
        mk = lambda dpr: vits.__dict__[args.arch](patch_size=8, drop_path_rate=dpr)
    student = MultiCropWrapper(mk(0.1), vits.DINOHead(dims[0], K, use_bn=False, norm_last_layer=True))
    teacher = MultiCropWrapper(mk(0.0), vits.DINOHead(dims[0], K, use_bn=False))
    teacher.load_state_dict(student.state_dict())
    crit = DINOLoss(K, 2 + nl, 0.07, 0.04, 30, 41)
    eng = LafsPretrainEngine(student, teacher, crit, B, n_local=nl, clip_grad=3.0, freeze_last_layer=1,
                             use_graph=not args.no_graph, device=device)
    g = torch.Generator(device=device).manual_seed(rank)
    eng.in_global_all.copy_(torch.randn(eng.in_global_all.shape, device=device, generator=g).clamp_(-1, 1))
    eng.in_local_all.copy_(torch.randn(eng.in_local_all.shape, device=device, generator=g).clamp_(-1, 1))

    fe, views = None, None
    if args.frontend:
        from lafs_cvpr2024_amd.face_pre_pro.ViT_face import face_landmark_4simmin_glo_loc
        from lafs_cvpr2024_amd.landmark_frontend import LandmarkFrontEnd
        cnn = face_landmark_4simmin_glo_loc(loss_type='None', GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=768,
                                            depth=12, heads=11, mlp_dim=2048)
        impl = os.environ.get("LAFS_FRONTEND_CNN", "hip")        # hip | torch | torch_bf16
        fe = LandmarkFrontEnd(cnn, B, n_local=nl, device=device, cnn_impl="hip" if impl == "hip" else "torch",
                              cnn_dtype=torch.bfloat16 if impl == "torch_bf16" else torch.float32)
        views = torch.randn(2 * (2 + nl), B, 3, 112, 112, device=device, generator=g).clamp_(-1, 1)
        augm, u8 = None, None
        if args.augment:
            from lafs_cvpr2024_amd.augment import DeviceAugmenter
            augm = DeviceAugmenter(B, n_local=nl, device=device, seed=rank)
            u8 = torch.randint(0, 256, (B, 3, 112, 112), device=device, dtype=torch.uint8, generator=g)
            views = augm(u8)
        fe.prefetch(views)

    niter = 1000
    lr_s = cosine_scheduler(5e-4 * B * world / 256., 1e-6, 41, niter, warmup_epochs=10)
    wd_s = cosine_scheduler(0.04, 0.4, 41, niter)
    mom_s = cosine_scheduler(0.996, 1, 41, niter)
    epoch = 1
    tt = float(crit.teacher_temp_schedule[epoch])

    def one(it):
        if fe is None:
            return eng.step(lr=float(lr_s[it]), wd=float(wd_s[it]), momentum=float(mom_s[it]), teacher_temp=tt, epoch=epoch)
        nxt = augm(u8) if augm is not None else views            # next batch: uint8 images -> 20 views (one launch)
        ev = torch.cuda.current_stream().record_event()          # "next batch's views are ready"
        fe.commit(eng)
        loss = eng.step(lr=float(lr_s[it]), wd=float(wd_s[it]), momentum=float(mom_s[it]), teacher_temp=tt, epoch=epoch)
        fe.prefetch(nxt, produced=ev)                            # front-end of the next batch overlaps this step
        return loss

    it0 = epoch * niter
    for i in range(args.warmup):
        one(it0 + i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = one(it0 + args.warmup + i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_v = float(loss.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        crops = world * B * (2 + nl)
        fl = step_flops(dims, B, nl, K)
        out = {
            "metric": "face-crops/sec/node (ViT-S LAFS pretrain, 2g+8l crops)", "value": round(crops / (ms * 1e-3), 1),
            "unit": "face-crops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.arch}/8{' (depth %d)' % args.depth if args.depth > 0 else ''} LAFS pretrain step, 2 global 112x112 + {nl} local 48x48 crops, "
                                   f"batch {B}/GPU, out_dim {K}, drop_path 0.1, dp{world}",
                       "global_batch": B * world, "parallelism": f"dp{world}", "hip_graph": not args.no_graph,
                       "landmark_frontend_in_step": bool(args.frontend), "device_augmentation_in_step": bool(args.augment)},
            "images_per_s": round(world * B / (ms * 1e-3), 1),
            "step_tflops_per_gpu": round(fl / (ms * 1e-3) / 1e12, 1),
            "step_mfma_frac": round(fl / (ms * 1e-3) / 2.5e15, 4),
            "final_loss": round(loss_v, 4), "debug_flags": dbg,
            "comm": {"backend": (dist.get_backend() if world > 1 else None), "ranks_seen": comm_ranks},
            # routing / scheduling knobs of the library and the engine that were set in the environment (none of them changes results;
            # an empty object = the shipped configuration)
            "lafs_env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("LAFS_") and k != "LAFS_BENCH_SHARE_GPU"},
        }
        hb = step_hbm_bytes()
        if hb is not None:                           # measured HBM bytes of one step (committed PMC profile) over this run's time
            out["step_hbm_bytes"] = hb
            out["step_hbm_frac"] = round(hb / (ms * 1e-3) / 8e12, 4)
        if not args.no_roofline:
            out["roofline"] = dominant_kernel_roofline(device)
        out["profile_provenance"] = PROFILE_NOTES       # which committed profile each read-back number came from (null = stale)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dims, nl, K, args.cpu_batch)
        if world == 1 and not args.no_extras and args.arch == "vit_small" and not args.frontend:
            del eng, student, teacher, crit
            out["extras"] = run_extras(device)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
