"""Byte-floor model of the C2 step (DESIGN.md section 6, round 5's kill criterion): per kernel class, the ALGORITHMIC bytes of this build's
dataflow (every operand read once, every result written once; weights and per-row statistics left out: they stay in the L2 / are < 1 %)
divided by the copy rate measured on this chip (profiles/lab_hbm_per_cu.txt: 5.2-5.9 TB/s for a read + write stream), beside the kernel
time of the newest serialised profile (profiles/round*_serial_kernel_stats.csv) when one is there.

    python tools/byte_floor.py            # table for BASELINE.json's C2 (ViT-S/8, 2 x 112 px + 8 x 48 px crops, batch 64, K = 100 000)
    python tools/byte_floor.py --round5   # the dataflow of rounds 1-5 (every operator a kernel)

Round 6 (default): the block's MLP is one launch in each direction (csrc/mlp_fused.hip) with LayerNorm 2 as its prologue -- the fc2 read of
gelu(u), the fc1-input-gradient read of du, the teacher's gelu(u) altogether and the LayerNorm-2 output's round trip (teacher: write +
read; student: the read) are gone.

Bytes are counted in units of D x T (channels x token rows) per layer; bf16 = 2, fp32 = 4 bytes per element:
  forward   LN (4 in, 2 out) | qkv GEMM (2 in, 6 out) | attention (6 in, 2 out) | projection + residual (2 + 4 in, 4 out) | LN | fc1 (2 in,
            8 gelu'(u) + 8 gelu(u) out; the teacher writes gelu(u) only) | fc2 + residual (8 + 4 in, 4 out)
  backward  GELU' input gradient (2 + 8 in, 8 out) | fc1 input gradient (8 in, 2 out) | LN backward (2 + 4 + 4 in, 4 + 2 out) | projection
            input gradient (2 in, 2 out) | attention backward (6 + 2 + 2 in, 6 out) | qkv input gradient (6 in, 2 out) | LN backward |
            four weight gradients (2 + 6, 2 + 2, 2 + 8, 8 + 2 in)
"""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D, DEPTH, MLP = 384, 12, 4
B, NG, NL, TG, TL = 64, 2, 8, 197, 37
K_OUT, BOTTLE, HIDDEN = 100000, 256, 2048
COPY_LO, COPY_HI = 5.2e12, 5.9e12

T_s = B * (NG * TG + NL * TL)           # 44 160 student token rows
T_t = B * NG * TG                        # 25 216 teacher token rows
fwd = {"LayerNorm forward (x2)": 2 * (4 + 2), "qkv GEMM": 2 + 6, "attention forward": 6 + 2, "projection + residual": 2 + 4 + 4,
       "fc1 + GELU (K = 384)": 2 + 2 * 2 * MLP, "fc2 + residual": 2 * MLP + 4 + 4}
fwd_t = dict(fwd, **{"fc1 + GELU (K = 384)": 2 + 2 * MLP})

bwd = {"GELU' input gradient": 2 + 2 * MLP + 2 * MLP, "fc1 input gradient": 2 * MLP + 2, "LayerNorm backward (x2)": 2 * (2 + 4 + 4 + 4 + 2),
       "projection input gradient": 2 + 2, "attention backward": 6 + 2 + 2 + 6, "qkv input gradient": 6 + 2,
       "weight gradients (4 per block)": (2 + 6) + (2 + 2) + (2 + 2 * MLP) + (2 * MLP + 2)}
ROUND5 = "--round5" in sys.argv
if not ROUND5:
    # forward: LN 1 (4 in, 2 out) | qkv | attention | projection | fused MLP: residual stream in (4: LayerNorm input AND residual) and out (4),
    #          student also h2 (2), gelu'(u) and gelu(u) (2 x 2 MLP)
    #          LN 1: block 0 is a launch (4 in, 2 out); blocks 1.. are written by the previous block's fused MLP from registers (2 out)
    fwd = {"LayerNorm forward (LN 1)": (4 + 2 + 2 * (DEPTH - 1)) / DEPTH, "qkv GEMM": 2 + 6, "attention forward": 6 + 2, "projection + residual": 2 + 4 + 4,
           "fused MLP forward (LN 2 + fc1 + GELU + fc2 + residual)": 4 + 4 + 2 + 2 * 2 * MLP}
    fwd_t = dict(fwd, **{"fused MLP forward (LN 2 + fc1 + GELU + fc2 + residual)": 4 + 4})
    # backward: fused MLP (dY 2 + gelu' 2 MLP in; du 2 MLP + dX 2 out) | the rest as before
    bwd = {"fused MLP backward (GELU' + fc1 input gradients)": 2 + 2 * MLP + 2 * MLP + 2, "LayerNorm backward (x2)": 2 * (2 + 4 + 4 + 4 + 2),
           "projection input gradient": 2 + 2, "attention backward": 6 + 2 + 2 + 6, "qkv input gradient": 6 + 2,
           "weight gradients (4 per block)": (2 + 6) + (2 + 2) + (2 + 2 * MLP) + (2 * MLP + 2)}

rows = []
for name in fwd:
    rows.append((name, DEPTH * D * (fwd[name] * T_s + fwd_t[name] * T_t)))
for name in bwd:
    rows.append((name, DEPTH * D * bwd[name] * T_s))
n_trunk = DEPTH * (4 * D * D + 2 * MLP * D * D) + 64 * 3 * D + 2 * D
n_head = D * HIDDEN + HIDDEN * HIDDEN + HIDDEN * BOTTLE + BOTTLE * K_OUT
# AdamW + EMA per parameter: read master, grad, exp_avg, exp_avg_sq, teacher (20 B); write master, both moments, two bf16 shadows, teacher (+ its shadow) (22 B)
rows.append(("AdamW + clip + EMA (%.1f M parameters)" % ((n_trunk + n_head) / 1e6), (n_trunk + n_head) * 42))
# DINO head: the last layer's weight-normed matrix (bf16, student + teacher) read by forward and gradient kernels, its gradient written;
# logits are not stored (fused head + loss); crops in, patch embedding
rows.append(("DINO head last layer (K = 100 000) + loss", BOTTLE * K_OUT * 2 * 4 + BOTTLE * K_OUT * 4))
rows.append(("crops -> patch tokens", B * 3 * (NG * 112 * 112 + NL * 48 * 48) * 4 * 2 + D * (T_s + T_t) * 4))

classes6 = {"LayerNorm forward (LN 1)": r"ln_fwd", "qkv GEMM": None, "attention forward": r"attn_fwd", "projection + residual": None,
            "fused MLP forward (LN 2 + fc1 + GELU + fc2 + residual)": r"mlp_fused_kernel<[01]", "fused MLP backward (GELU' + fc1 input gradients)": r"mlp_fused_kernel<2",
            "LayerNorm backward (x2)": r"ln_bwd|ln_fold", "projection input gradient": None, "attention backward": r"attn_bwd",
            "qkv input gradient": None, "weight gradients (4 per block)": r"wgrad"}
classes = {  # kernel-name patterns of the serialised profile per model row (student + teacher launches together)
    "LayerNorm forward (x2)": r"ln_fwd", "qkv GEMM": None, "attention forward": r"attn_fwd", "projection + residual": None,
    "fc1 + GELU (K = 384)": None, "fc2 + residual": None, "GELU' input gradient": None, "fc1 input gradient": None,
    "LayerNorm backward (x2)": r"ln_bwd|ln_fold", "projection input gradient": None, "attention backward": r"attn_bwd",
    "qkv input gradient": None, "weight gradients (4 per block)": r"wgrad", }
if not ROUND5:
    classes = classes6
measured, src = {}, None
files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_serial_kernel_stats.csv")))
if files:
    src = files[-1]
    stats = list(csv.DictReader(open(src)))
    head = open(src.replace("_stats.csv", "_table.txt")).read() if os.path.exists(src.replace("_stats.csv", "_table.txt")) else ""
    m = re.search(r"\((\d+) steps\)", head)
    steps = int(m.group(1)) if m else 13
    for name, pat in classes.items():
        if pat:
            measured[name] = sum(int(r["TotalDurationNs"]) for r in stats if re.search(pat, r["Name"])) / steps / 1e6
    measured["all GEMMs (K-resident + tiled)"] = sum(int(r["TotalDurationNs"]) for r in stats if re.search(r"gemm_kres|gemm_nt_kernel|gemm_big", r["Name"])) / steps / 1e6

tot = sum(b for _, b in rows)
print("C2 step, algorithmic bytes of this dataflow (student %d rows, teacher %d rows, D = %d, %d blocks)" % (T_s, T_t, D, DEPTH))
print("%-46s %9s %8s %16s %s" % ("kernel class", "GB", "share", "ms at copy rate", "measured ms (serialised)" if src else ""))
gemm_rows = [n for n in classes if classes[n] is None]
for name, b in rows:
    t_lo, t_hi = b / COPY_HI * 1e3, b / COPY_LO * 1e3
    extra = "%.2f" % measured[name] if name in measured else ""
    print("%-46s %9.2f %7.1f%% %7.2f - %5.2f   %s" % (name, b / 1e9, 100.0 * b / tot, t_lo, t_hi, extra))
gb = sum(b for n, b in rows if n in gemm_rows)
print("%-46s %9.2f %7.1f%% %7.2f - %5.2f   %s" % ("  (the GEMM rows without a pattern together)", gb / 1e9, 100.0 * gb / tot, gb / COPY_HI * 1e3, gb / COPY_LO * 1e3,
                                                 "%.2f" % measured["all GEMMs (K-resident + tiled)"] if measured else ""))
print("%-46s %9.2f %7s  %7.2f - %5.2f" % ("TOTAL (floor of this dataflow)", tot / 1e9, "", tot / COPY_HI * 1e3, tot / COPY_LO * 1e3))
if src:
    print("measured column: %s" % os.path.relpath(src, ROOT))
# exact byte cuts that stay inside this dataflow (sized)
DT = D * T_s * DEPTH
print("exact cuts sized in DESIGN section 6: LayerNorm backward in the N = 384 input-gradient epilogues -%.2f GB; one stored pre-activation "
      "-%.2f GB; weight-gradient over-fetch (fabric bytes, not algorithmic) -1.8 GB" % (8 * DT / 1e9, 2 * MLP * DT / 1e9))
