#!/usr/bin/env python3
"""Per-workgroup phase timeline of one NT GEMM launch (ablation build: LAFS_USE_ABLATE_LIB=1).  Each workgroup stamps
s_memtime at start, end of the k-loop, last store issued, stores acknowledged, plus HW_ID / XCC_ID; this prints phase statistics
and, for a few CUs, who overlapped whom.   usage: LAFS_USE_ABLATE_LIB=1 python tools/lab/nt_timeline.py [fc1|dgelu|qkv|fc2]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from lafs_cvpr2024_amd import _lib, ops

dev, bf, T = "cuda", torch.bfloat16, 44160
which = (sys.argv[1:] or ["fc1"])[0]
shape = {"fc1": (T, 1536, 384, _lib.EPI_BF16_GELU), "dgelu": (T, 1536, 384, _lib.EPI_DGELU_BF16), "qkv": (T, 1152, 384, _lib.EPI_BF16),
         "fc2": (T, 384, 1536, _lib.EPI_RESID_F32), "fc1d": (T, 384, 1536, _lib.EPI_BF16)}[which]
M, N, K, epi = shape
A = torch.randn(M, K, device=dev).to(bf); B = (torch.randn(N, K, device=dev) * .02).to(bf)
f32 = epi in (_lib.EPI_RESID_F32, _lib.EPI_F32)
out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else bf)
kw = {}
if epi == _lib.EPI_BF16_GELU: kw["out2"] = torch.empty(M, N, device=dev, dtype=bf)
if epi == _lib.EPI_RESID_F32: kw["resid"] = torch.randn(M, N, device=dev)
if epi == _lib.EPI_DGELU_BF16: kw["aux"] = torch.randn(M, N, device=dev).to(bf)
bias = None if epi == _lib.EPI_DGELU_BF16 else torch.zeros(N, device=dev)
run = lambda: ops.gemm_nt(A, B, epi, bias=bias, out=out, **kw)
for _ in range(3): run()
h = _lib.lib()
h.lafs_lab_set_stamps.argtypes = [C.c_void_p]; h.lafs_lab_set_stamps.restype = None
n_wg = 8192
st = torch.zeros(n_wg, 8, dtype=torch.int64, device=dev)
h.lafs_lab_set_stamps(st.data_ptr())
run(); torch.cuda.synchronize()
h.lafs_lab_set_stamps(None)
s = st.cpu().numpy()
s = s[s[:, 0] != 0]
for _ in range(2): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
wall_us = e0.elapsed_time(e1) * 1e3
xcc = s[:, 5] & 0xF
rel = np.zeros((len(s), 4)); spans = []
for x in np.unique(xcc):                              # every XCD has its own s_memtime counter: times are relative to the XCD's first stamp
    sel = xcc == x
    t0 = s[sel, 0].min()
    rel[sel] = (s[sel, :4] - t0).astype(np.float64)
    spans.append(rel[sel, 3].max())
clk = np.mean(spans) / (wall_us * 1e-6)           # tick rate, calibrated on the launch's wall time
rel = rel / clk * 1e6
span = rel[:, 3].max()
print(f"{which}: {len(s)} tiles, launch {wall_us:.1f} us, s_memtime at {clk/1e9:.2f} GHz")
main, epi_issue, drain = rel[:, 1] - rel[:, 0], rel[:, 2] - rel[:, 1], rel[:, 3] - rel[:, 2]
for nm, v in (("k-loop", main), ("epilogue issue", epi_issue), ("store drain", drain), ("whole workgroup", rel[:, 3] - rel[:, 0])):
    print(f"  {nm:16s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f} us")
hw = s[:, 4]
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 16 + cu
ks, counts = np.unique(key, return_counts=True)
print(f"  distinct (xcc, se, sh, cu) = {len(ks)}; workgroups per CU: min {counts.min()} max {counts.max()}")
for k in ks[:3]:
    rows = rel[key == k]; rows = rows[np.argsort(rows[:, 0])]
    print(f"  CU {k}:")
    for r in rows:
        print(f"     start {r[0]:7.2f}  k-loop end {r[1]:7.2f}  stores issued {r[2]:7.2f}  acked {r[3]:7.2f}")
# how much of the launch does a CU spend with 0 / 1 / 2 workgroups in their k-loop, and in the epilogue?
grid = np.linspace(0, span, 2000)
inloop = np.zeros((len(ks), len(grid))); inepi = np.zeros_like(inloop)
for i, k in enumerate(ks):
    for r in rel[key == k]:
        inloop[i] += (grid >= r[0]) & (grid < r[1]); inepi[i] += (grid >= r[1]) & (grid < r[3])
for nm, a in (("k-loop", inloop), ("epilogue+drain", inepi)):
    print(f"  fraction of CU-time with 0/1/2+ workgroups in {nm}: " + " ".join(f"{(a == c).mean():.2f}" for c in (0, 1)) + f" {(a >= 2).mean():.2f}")
