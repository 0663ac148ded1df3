#!/usr/bin/env python3
"""Phase spans of ONE graph-replayed step from a rocprofv3 --kernel-trace CSV (GPU box or container):
forward (first kernel .. DINO loss), loss + head backward, trunk backward, update tail; per phase the GPU-busy time summed over
kernels (serialised cost) against the wall span -- i.e. how much the concurrent streams buy and where the step's time sits.
usage: tools/step_timeline.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(anonymous namespace\)::|void |\(.*", "", r["Kernel_Name"])
marks = [i for i, r in enumerate(rows) if "zero_chunks" in r["Kernel_Name"]]
if len(marks) < 3:
    sys.exit("need at least 3 steps in the trace")
step = rows[marks[-2]:marks[-1]]
t0 = int(step[0]["Start_Timestamp"])
S = lambda r: (int(r["Start_Timestamp"]) - t0) / 1e3
E = lambda r: (int(r["End_Timestamp"]) - t0) / 1e3
first = lambda pat: next((r for r in step if pat in r["Kernel_Name"]), None)
last = lambda pat: next((r for r in reversed(step) if pat in r["Kernel_Name"]), None)
loss0 = first("row_stats") or first("head_stats")
bwd0 = first("scatter_cls")
upd0 = first("center_ema")
end = max(E(r) for r in step)
cuts = [("forward (both networks, heads)", 0.0, S(loss0)), ("DINO loss + head backward", S(loss0), S(bwd0)),
        ("trunk backward", S(bwd0), S(upd0)), ("update tail", S(upd0), end)]
print(f"step: {len(step)} launches, {end / 1e3:.3f} ms from first start to last end")
for title, a, b in cuts:
    ks = [r for r in step if a <= S(r) < b]
    busy = sum(E(r) - S(r) for r in ks)
    print(f"  {title:34s} span {(b - a) / 1e3:7.3f} ms   kernels {len(ks):4d}   summed kernel time {busy / 1e3:7.3f} ms   overlap factor {busy / max(b - a, 1e-9):4.2f}")
    by = collections.Counter()
    for r in ks:
        by[name(r)[:70]] += E(r) - S(r)
    for k, v in by.most_common(7):
        print(f"        {v / 1e3:7.3f} ms  {k}")
# idle gaps: time with no kernel running
ev = sorted([(S(r), 1) for r in step] + [(E(r), -1) for r in step])
run, idle, prev = 0, 0.0, 0.0
for t, d in ev:
    if run == 0:
        idle += t - prev
    run += d
    prev = t
print(f"  GPU idle inside the step (no kernel running): {idle / 1e3:.3f} ms")
