#!/bin/bash
# final evidence of round 5, ONE gpurun call: everything profiles/round5_* is condensed from
R=$GRAFT_REPO_ROOT; cd $R
bash tools/collect_profiles_round.sh > gpurun_out/collect.log 2>&1; tail -30 gpurun_out/collect.log | cut -c1-400
mkdir -p gpurun_out/r5
bash tools/profile_extras_r5.sh mynet > gpurun_out/r5/mynet.log 2>&1; tail -5 gpurun_out/r5/mynet.log | cut -c1-200
bash tools/profile_extras_r5.sh finetune > gpurun_out/r5/finetune.log 2>&1; tail -5 gpurun_out/r5/finetune.log | cut -c1-200
# the driver's own sequence on the same box: default bench run (headline + extras + cpu baseline)
cd $R; timeout 1500 python bench.py > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err; tail -c 3000 gpurun_out/r5/bench_default.json
