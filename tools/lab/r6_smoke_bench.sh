#!/bin/bash
# round 6: smoke + the default bench run (committed as profiles/round6_bench_default_run.json); the GPU suite ran in r6_tests_collect.sh
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 > gpurun_out/r6_final_smoke.txt
timeout 1200 python bench.py > gpurun_out/r6_bench_default_run.json 2> gpurun_out/r6_bench_default_run.err
cat gpurun_out/r6_final_smoke.txt; cut -c1-900 gpurun_out/r6_bench_default_run.json
