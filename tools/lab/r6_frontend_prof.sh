#!/bin/bash
# round 6 lab: kernel stats of the headline step WITH the landmark front-end in it (bench.py --frontend)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/frontend; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o fe --output-format csv -- python3 $R/bench.py --frontend --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > $O/out.json 2> $O/err.txt
python3 - <<PY
import csv,glob,re
f=glob.glob("$O/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
vit=re.compile(r"mlp_fused|gemm_kres|wgrad|ln_bwd|ln_fwd|attn_|clip_adamw|head_|weightnorm|chunk_sumsq|seg_|ln_fold|l2norm|embed|pos_|patchify|zero_chunks|fill_zero|droppath|gather_cls|center_ema|transpose_cast|scale_cast")
tot=0
for r in rows:
    n=re.sub(r"\(anonymous namespace\)::|void ","",r["Name"])
    if vit.search(n): continue
    ms=int(r['TotalDurationNs'])/13e6; tot+=ms
    if ms>0.01: print(f"{n[:90]:90s} {int(r['Calls']):6d} {ms:8.3f} ms/step {float(r['AverageNs'])/1e3:9.1f} us")
print('non-ViT kernels total ms/step', tot)
PY
cut -c1-300 $O/out.json
