#!/bin/bash
# GPU box, one gpurun call: whole-step A/B of this round's switches on ONE box (each line: ms_per_step of bench.py, 30 steps).
# usage: bash tools/lab/r3_ab.sh "<ENV1=.. ENV2=..>" "<...>" ...   (each argument = one environment; "" = shipped configuration)
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/ab; mkdir -p $O
i=0
for envs in "$@"; do
  i=$((i+1))
  for rep in 1 2; do
    ms=$(env $envs python3 $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras 2>$O/err_$i.txt | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null)
    echo "[ab] '$envs' run $rep: ${ms:-FAILED} ms/step" | tee -a $O/ab.txt
  done
done
