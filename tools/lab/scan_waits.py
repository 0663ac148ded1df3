"""Compile every csrc/*.hip to gfx950 assembly and list the kernels whose global stores / loads are each followed or preceded by
`s_waitcnt vmcnt(0)` -- the signature of a guarded epilogue (every `if (in_range) store` is a basic block of its own, and hipcc opens a
block that touches registers written by earlier loads with a full wait, which also sits out every store in flight).  Round 5 found
the plain epilogue of gemm_big.hip (22 serialised stores per tile), the tiled GEMM's residual epilogue (its one-row-ahead prefetch was
waited for where it was issued) and the weight-gradient kernel (144 serialised stores per wave) this way.
usage: python tools/lab/scan_waits.py [min_pairs]"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "lafs_cvpr2024_amd", "csrc")
os.makedirs("/tmp/asm", exist_ok=True)
procs = []
for f in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
    out = "/tmp/asm/" + os.path.basename(f)[:-4] + ".s"
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(f):
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"),
                                       "-I" + SRC, "-Wno-unused-function", "-S", "--cuda-device-only", f, "-o", out], stderr=subprocess.DEVNULL))
for p in procs:
    p.wait()
def kind(t):
    m = t.split()[0] if t.split() else ""
    if m.startswith("global_load_lds"): return "DMA"
    if m.startswith(("global_load", "buffer_load")): return "GL"
    if m.startswith(("global_store", "buffer_store", "global_atomic")): return "GS"
    if "vmcnt(0)" in t: return "W0"
    return None
floor = int(sys.argv[1]) if len(sys.argv) > 1 else 8
res = []
for f in sorted(glob.glob("/tmp/asm/*.s")):
    s = open(f).read()
    for m in re.finditer(r"^(_Z[A-Za-z0-9_]+):", s, re.M):
        a, b = m.end(), s.find(".Lfunc_end", m.end())
        if b < 0: continue
        ev = [k for k in (kind(l.strip()) for l in s[a:b].split("\n")) if k]
        ws = sum(1 for i in range(len(ev) - 1) if ev[i] == "W0" and ev[i + 1] == "GS")
        lw = sum(1 for i in range(len(ev) - 1) if ev[i] == "GL" and ev[i + 1] == "W0")
        if ws >= floor or lw >= floor:
            res.append((os.path.basename(f), m.group(1), ws, ev.count("GS"), lw, ev.count("GL")))
names = subprocess.run(["c++filt"], input="\n".join(r[1] for r in res), capture_output=True, text=True).stdout.split("\n")
for r, nm in zip(res, names):
    print(f"{r[0]:18s} wait->store {r[2]:3d} of {r[3]:3d}   load->wait {r[4]:3d} of {r[5]:3d}   {re.sub(r'.anonymous namespace.::', '', nm)[:110]}")
