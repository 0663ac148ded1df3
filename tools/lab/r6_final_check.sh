#!/bin/bash
# round 6: the round-end check the driver runs -- GPU suite, smoke, default bench (committed as profiles/round6_bench_default_run.json)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) > gpurun_out/r6_final_tests.txt 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 > gpurun_out/r6_final_smoke.txt
timeout 1200 python bench.py > gpurun_out/r6_bench_default_run.json 2> gpurun_out/r6_bench_default_run.err
cat gpurun_out/r6_final_tests.txt gpurun_out/r6_final_smoke.txt; cut -c1-900 gpurun_out/r6_bench_default_run.json
